#!/usr/bin/env python3
"""Do the batches of the fixed-base (and variable-base) workload gain from two of them in flight?  k_normalize runs one wave
per SIMD (a chain of dependent instructions per lane); the main kernel of the NEXT batch could issue beside it.  Forecast with
two contexts (one stream and one set of scratch buffers each) taking the queued batches in turn, against one context taking all
of them — the same K batches, drained inside the timed region.     python tools/gpu_fixed_overlap.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ecgpu = importlib.import_module("elliptic-curves_amd")
dev = torch.device("cuda:0")
N = 1 << 20
K = 24


def run(label, cid, L, engines, var):
    g = torch.Generator(device="cpu").manual_seed(5)
    scal = torch.randint(0, 256, (N, L), dtype=torch.uint8, generator=g)
    scal[:, 0] &= 0x7F
    d_scal = scal.to(dev)
    outs = [(torch.empty((N, 2 * L), dtype=torch.uint8, device=dev), torch.empty((N,), dtype=torch.uint8, device=dev)) for _ in engines]
    d_pts = None
    if var:
        engines[0].mul_by_generator_dev(cid, d_scal, N, *outs[0])
        engines[0].synchronize()
        d_pts = outs[0][0].clone()
    torch.cuda.synchronize()

    def step(i):
        e = engines[i % len(engines)]
        o = outs[i % len(engines)]
        if var:
            e.mul_dev(cid, d_scal, d_pts, None, N, *o)
        else:
            e.mul_by_generator_dev(cid, d_scal, N, *o)

    for i in range(2 * len(engines)):
        step(i)
    for e in engines:
        e.synchronize()
        e.set_async(True)
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        for e in engines:
            e.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        best = dt if best is None else min(best, dt)
    for e in engines:
        e.set_async(False)
    print("%-34s %d in flight: %.4f ms per batch" % (label, len(engines), best), flush=True)
    return outs


a, b = ecgpu.Engine(), ecgpu.Engine()
if os.environ.get("FIXED_TIMING") == "0":          # without the per-call timing events (ecgpu_set_timing)
    a.set_timing(False)
    b.set_timing(False)
for name, cid, L, var in (("k256 fixed base", ecgpu.K256, 32, False), ("p256 variable base", ecgpu.P256, 32, True),
                          ("p384 variable base", ecgpu.P384, 48, True)):
    o1 = run(name, cid, L, [a], var)
    o2 = run(name, cid, L, [a, b], var)
    assert torch.equal(o1[0][0], o2[0][0]) and torch.equal(o2[0][0], o2[1][0])
    o1 = run(name, cid, L, [a], var)
    o2 = run(name, cid, L, [a, b], var)
