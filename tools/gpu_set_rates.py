#!/usr/bin/env python3
"""Rates of one parameter set (2^20 units per call, device-resident inputs): fixed base, variable base, MSM.
    python tools/gpu_set_rates.py <name> [<name> ...]       names: k256 p256 p384 sm2 p224 p192 p521 bp256 bp384 bp256t1 bp384t1 bign256
ECGPU_TOOL_LIB=<lib/libecgpu_<suffix>.so of tools/build_alt_lib.sh> measures that build instead (before / after comparisons on one box)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
from gpu_common import rand_scalars  # noqa: E402

    print("library: %s" % ec.LIB_PATH)

e = ec.Engine(0)
n = 1 << 20
for name in sys.argv[1:] or ["p521"]:
    cid = ec.CURVE_IDS[name]
    L = ec.FIELD_BYTES[cid]
    d_k = e.to_device(rand_scalars(cid, n, 0x5E7 + cid))
    d_k2 = e.to_device(rand_scalars(cid, n, 0x5E8 + cid))
    d_p, d_o, d_f = e.dev_alloc(n * 2 * L), e.dev_alloc(n * 2 * L), e.dev_alloc(n)
    for rep in range(3):
        e.mul_by_generator_dev(cid, d_k, n, d_p, d_f)
    fb, fbt = e.last_timing("main"), e.last_timing("total")
    for rep in range(2):
        e.mul_dev(cid, d_k2, d_p, None, n, d_o, d_f)
    vb = e.last_timing("main")
    for rep in range(3):
        t0 = time.perf_counter()
        e.lincomb_dev(cid, d_k2, d_p, None, n, d_o, d_f)
        ms = (time.perf_counter() - t0) * 1e3
    print("%-8s fixed base: kernel %.3f ms, call %.3f ms -> %.3e /s | variable base: kernel %.3f ms -> %.3e /s | MSM 2^20: %.2f ms -> %.3e terms/s (accumulate %.2f)" % (
        name, fb, fbt, n / fbt * 1e3, vb, n / vb * 1e3, ms, n / ms * 1e3, e.last_timing("accumulate") or 0), flush=True)
    for b in (d_k, d_k2, d_p, d_o, d_f):
        b.free()
e.close()
