#!/usr/bin/env python3
"""Degenerate scalar sets through the 2^24-term k256 MSM (device-resident): every term of a window in ONE bucket (all scalars
equal), in two buckets (every other scalar equal), all ones — against random scalars.  The accumulation lanes own chunks, not
buckets, so these cost what random input costs there; the sort's level B hands a whole window to ONE workgroup (its lanes rank
with one LDS atomic per distinct key and wave), which is what this measures.  Every result is checked: sum_i k P_i = k sum_i P_i.
    python tools/gpu_msm_degenerate.py [log2 terms]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
from gpu_common import rand_scalars  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
cid, L = 0, 32
e = ec.Engine(0)
d_s = e.to_device(rand_scalars(cid, n, 0xDE6E))
d_p, d_f = e.dev_alloc(n * 2 * L), e.dev_alloc(n)
e.mul_by_generator_dev(cid, d_s, n, d_p, d_f)
d_o, d_i = e.dev_alloc(2 * L + 64), e.dev_alloc(16)
d_t, d_ti = e.dev_alloc(2 * L + 64), e.dev_alloc(16)
e.point_sum_dev(cid, d_p, None, n, d_t, d_ti)
total = e.to_host(d_t, 2 * L)
k0 = bytes(rand_scalars(cid, 1, 0xDE6F))
k1 = bytes(rand_scalars(cid, 1, 0xDE70))
one = (1).to_bytes(L, "big")
cases = [("random", None), ("all scalars equal", [k0]), ("every other scalar equal", [k0, k1]), ("all ones", [one])]
for name, pat in cases:
    if pat is None:
        ks = rand_scalars(cid, n, 0xDE71)
    else:
        ks = np.tile(np.frombuffer(b"".join(pat), np.uint8), n // len(pat))
    d_k = e.to_device(ks)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        e.lincomb_dev(cid, d_k, d_p, None, n, d_o, d_i)
        ts.append(time.perf_counter() - t0)
    stages = {s: round(e.last_timing(s), 3) for s in ("sort", "accumulate", "reduce")}
    ok = "-"
    if pat is not None and len(pat) == 1:
        w, wf = e.mul(cid, pat[0], total)
        ok = bytes(e.to_host(d_o, 2 * L)) == bytes(w)
    print("2^%d terms, %-26s %8.3f ms  %s  check: %s" % (lg, name + ":", min(ts[1:]) * 1e3, stages, ok), flush=True)
    d_k.free()
e.close()
