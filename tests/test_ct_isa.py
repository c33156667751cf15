"""The uniform-schedule kernels (csrc/ecgpu_ct.h) on their gfx950 ISA: tools/ct_isa_check.py compiles the translation unit
to assembly (no GPU needed) and runs a register-level taint analysis from every loaded record to every branch condition
and every memory address.  The same analysis must flag the variable-time kernels (self-test), otherwise it proves nothing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ct_isa_check.py")


def _run(*args):
    return subprocess.run([sys.executable, TOOL, *args], capture_output=True, text=True, timeout=1500)


# k256 / p256 (the BASELINE curves) and the other reduction families with code of their own: p384 (signed sparse rows) and
# bign256 (two-term rows, little-endian records, generic a).  `python tools/ct_isa_check.py --curve <X>Params` checks any
# set; profiles/r04/ct_isa_check.txt holds all twelve (p521 — Mersenne rows, 66-byte records — takes 25 s and is run there).
@pytest.mark.parametrize("curve", ["K256Params", "P256Params", "P384Params", "Bign256Params"])
def test_no_branch_or_address_depends_on_scalar_or_point_data(curve):
    r = _run("--curve", curve)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("-> OK") == 3, r.stdout
    assert all(k in r.stdout for k in ("k_var_base_ct", "k_fixed_base_ct", "k_proj_sum_level")), r.stdout


def test_checker_flags_the_variable_time_kernels():
    r = _run("--self-test", "--curve", "P192Params")            # (the smallest parameter set: the self-test is about the checker, not the curve)
    assert r.returncode == 0, r.stdout + r.stderr              # 0 = both variable-time kernels were reported
    # k_var_base<C, false>, k_var_base<C, true> (+ a G), k_fixed_base<C, false / true> (record- / quad-major hand-over to k_normalize)
    assert r.stdout.count("VIOLATIONS") == 4, r.stdout
    assert "load address from tainted register" in r.stdout and "branch on tainted" in r.stdout
