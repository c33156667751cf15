"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/ecgpu.h declares, and fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ecgpu():
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "elliptic-curves_amd", "lib", "libecgpu.so")):
        __graft_entry__.build()
    return importlib.import_module("elliptic-curves_amd")


def header_functions():
    src = open(os.path.join(ROOT, "include", "ecgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ecgpu_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(ecgpu):
    lib = ecgpu.load_library()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libecgpu.so does not export %s" % name
    assert sorted(ecgpu.ABI_SYMBOLS) == declared, "ABI_SYMBOLS out of sync with include/ecgpu.h"


def test_pure_queries_work_without_gpu(ecgpu):
    lib = ecgpu.load_library()
    assert lib.ecgpu_field_bytes(0) == 32 and lib.ecgpu_field_bytes(1) == 32 and lib.ecgpu_field_bytes(2) == 48
    assert lib.ecgpu_field_bytes(9) == 32 and lib.ecgpu_field_bytes(10) == 48 and lib.ecgpu_field_bytes(99) == 0
    assert b"gfx950" in lib.ecgpu_version()


def test_no_cpu_fallback(ecgpu):
    """On a machine without a gfx950 device the engine must refuse to start, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(ecgpu.EcgpuError) as e:
        ecgpu.Engine(0)
    assert e.value.code == ecgpu.ERR_NO_DEVICE
    ctx = ctypes.c_void_p()
    assert ecgpu.load_library().ecgpu_init(ctypes.byref(ctx), 0) == ecgpu.ERR_NO_DEVICE
    assert not ctx.value


def test_product_does_not_touch_oracle():
    """The product path must never import, link or call the oracle (test infrastructure)."""
    pkg = os.path.join(ROOT, "elliptic-curves_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "ecref" not in text and "oracle_lib" not in text and "liboracle" not in text, os.path.join(dirpath, f)
    out = os.popen("ldd %s" % os.path.join(pkg, "lib", "libecgpu.so")).read()
    assert "oracle" not in out
