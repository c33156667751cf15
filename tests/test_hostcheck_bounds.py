"""Re-runs the field / point / driver hostcheck tests against a -DECGPU_BOUNDS_CHECK build of the same code:
every lazily reduced element is checked at run time against the magnitude its type declares (the run-time
twin of the static_asserts in ecgpu_field.h; mirrors the reference's debug-build magnitude checker)."""
import ctypes
import fcntl
import os
import subprocess

import pytest

import hostcheck_lib as hc
import test_hostcheck as T

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "hostcheck", "libhostcheck_bounds.so")


@pytest.fixture(scope="module")
def bounds_lib():
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "elliptic-curves_amd", "csrc")
    # pytest-xdist workers arrive here together: one builds (temporary name, renamed when complete), the others wait
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
            tmp = "%s.%d.tmp" % (LIB, os.getpid())
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DECGPU_BOUNDS_CHECK", "-Wno-unknown-pragmas",
                                   "-o", tmp, src])
            os.replace(tmp, LIB)
    old = hc._lib
    hc._lib = ctypes.CDLL(LIB)
    yield
    hc._lib = old


@pytest.mark.parametrize("curve", ["k256", "p256", "p384", "sm2", "p224", "p192", "p521", "bp256", "bp384", "bp256t1", "bp384t1"])
def test_bounds_field_and_points(bounds_lib, oracle, curve):
    T.test_field_ops_vs_oracle_and_bigint(oracle, curve)
    T.test_field_lazy_chain(curve)
    T.test_point_ops_complete_formulas(oracle, curve)


@pytest.mark.parametrize("curve", ["k256", "p256", "p384", "sm2", "p224", "p192", "p521", "bp256", "bp384", "bp256t1", "bp384t1"])
def test_bounds_drivers(bounds_lib, oracle, curve):
    T.test_fixed_base_algorithm(oracle, curve, 8)
    T.test_var_base_algorithm(oracle, curve)
    T.test_var_base_ladder_corner_cases(oracle, curve)
    T.test_pippenger_algorithm(oracle, curve, 7)
    T.test_pippenger_skewed_scalars(oracle, curve)
