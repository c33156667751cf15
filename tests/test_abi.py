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


def test_group_entry_refuses_without_gpu(ecgpu):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(ecgpu.EcgpuError) as e:
        ecgpu.Group([0, 1])
    assert e.value.code == ecgpu.ERR_NO_DEVICE


def test_rust_binding_matches_header():
    """rustc is not in this image, so the reference-side binding is checked mechanically: every function include/ecgpu.h
    declares appears in elliptic-curves_amd/rust/ecgpu_sys.rs with the same name, arity, argument types and return type
    (C types mapped by tests/abi_parse.py), nothing else is declared there, and the enum constants agree.  The safe layer
    (ecgpu_shim.rs) may only call functions that exist."""
    import abi_parse
    header = abi_parse.parse_header(os.path.join(ROOT, "include", "ecgpu.h"))
    assert len(header) >= 50
    rust = abi_parse.parse_rust_extern(os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_sys.rs"))
    assert sorted(rust) == sorted(d[0] for d in header)
    for d in header:
        name, ret, args = abi_parse.rust_signature(d)
        assert rust[name] == (ret, args), "%s: header says %r, ecgpu_sys.rs says %r" % (name, (ret, args), rust[name])
    sys_src = open(os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_sys.rs")).read()
    hdr_src = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "ecgpu.h")).read(), flags=re.S)
    consts = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"(ECGPU_\w+)\s*=\s*(-?\d+)", hdr_src))
    assert len(consts) >= 19
    for k, v in consts.items():
        assert re.search(r"pub const %s: c_int = %d;" % (k, v), sys_src), k
    shim = open(os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_shim.rs")).read()
    shim_code = re.sub(r"//[^\n]*", "", shim)
    called = set(re.findall(r"\b(ecgpu_\w+)\s*\(", shim_code))
    assert called and called <= set(rust), called - set(rust)
    # and the generator reproduces the committed file (nobody edited it by hand)
    import subprocess, sys as _sys, tempfile, shutil
    with tempfile.TemporaryDirectory() as td:
        keep = os.path.join(td, "ecgpu_sys.rs")
        shutil.copy(os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_sys.rs"), keep)
        subprocess.check_call([_sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], stdout=subprocess.DEVNULL)
        assert open(keep).read() == open(os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_sys.rs")).read()
