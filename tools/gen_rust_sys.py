#!/usr/bin/env python3
"""Writes elliptic-curves_amd/rust/ecgpu_sys.rs — the raw `extern "C"` declarations of EVERY function include/ecgpu.h
declares, generated from the header (there is no rustc / bindgen in this image; tests/test_abi.py re-parses both files
and compares name, arity and types, so the two cannot drift apart).    python tools/gen_rust_sys.py"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi_parse  # noqa: E402

HEADER = os.path.join(ROOT, "include", "ecgpu.h")
OUT = os.path.join(ROOT, "elliptic-curves_amd", "rust", "ecgpu_sys.rs")


def enums(src):
    """enum constants of the header -> [(name, value)]"""
    out = []
    for blk in re.finditer(r"enum\s*\{(.*?)\}", re.sub(r"/\*.*?\*/", " ", src, flags=re.S), flags=re.S):
        for m in re.finditer(r"(ECGPU_\w+)\s*=\s*(-?\d+)", blk.group(1)):
            out.append((m.group(1), int(m.group(2))))
    return out


def main():
    decls = abi_parse.parse_header(HEADER)
    src = open(HEADER).read()
    lines = [
        "//! ecgpu_sys.rs — raw FFI declarations of libecgpu.so, GENERATED from include/ecgpu.h by tools/gen_rust_sys.py.",
        "//! Do not edit: regenerate.  tests/test_abi.py checks this file against the header (name, arity, types).",
        "//! The safe adapters behind the reference's traits are in ecgpu_shim.rs.",
        "#![allow(non_camel_case_types, dead_code)]",
        "",
        "use core::ffi::{c_char, c_int, c_void};",
        "",
        "/// `ecgpu_ctx`: one GPU, its stream, its device-resident tables (opaque).",
        "#[repr(C)]",
        "pub struct EcgpuCtx {",
        "    _private: [u8; 0],",
        "}",
        "/// `ecgpu_group`: one context per GPU of a node, driven from one process (opaque).",
        "#[repr(C)]",
        "pub struct EcgpuGroup {",
        "    _private: [u8; 0],",
        "}",
        "",
    ]
    for name, val in enums(src):
        lines.append("pub const %s: c_int = %d;" % (name, val))
    lines += ["", '#[link(name = "ecgpu")]', 'unsafe extern "C" {']
    for d in decls:
        name, ret, args = abi_parse.rust_signature(d)
        params = ", ".join("%s: %s" % (an if an != "type" else "ty", t) for (_, an), t in zip(d[2], args))
        sig = "    pub fn %s(%s)%s;" % (name, params, (" -> " + ret) if ret else "")
        if len(sig) > 118:                      # wrap long signatures
            sig = "    pub fn %s(\n        %s,\n    )%s;" % (name, ",\n        ".join("%s: %s" % (an, t) for (_, an), t in zip(d[2], args)),
                                                         (" -> " + ret) if ret else "")
        lines.append(sig)
    lines += ["}", ""]
    with open(OUT, "w") as f:
        f.write("\n".join(lines))
    print("wrote %s: %d functions" % (OUT, len(decls)))


if __name__ == "__main__":
    main()
