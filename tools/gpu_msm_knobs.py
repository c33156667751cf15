#!/usr/bin/env python3
"""Sweeps the MSM's tuning knobs (environment variables read at plan time) at the sizes that matter: one GPU's share of a
sharded 2^24-term MSM (2^21) and the whole thing (2^24).    python tools/gpu_msm_knobs.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0, variant="knobs")   # the tool build: the ECGPU_* knobs below are read there only (csrc/ecgpu_knobs.h)
e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda")
g.manual_seed(12)
nmax = 1 << 24
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g)
k[:, 0] &= 0x7F
pts = torch.empty((nmax, 64), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
e.mul_by_generator_dev(0, k, nmax, pts, None)
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g)
k[:, 0] &= 0x7F
r = torch.empty((1, 64), dtype=torch.uint8, device="cuda")
ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def timed(n, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.lincomb_dev(0, k[:n], pts[:n], None, n, r, ri)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = {s: e.last_timing(s) for s in ("sort", "accumulate", "reduce")}
    return min(ts[1:]) * 1e3, st


def sweep(n, var, values):
    base, st = timed(n)
    print("n=2^%d default %.3f ms %s" % (n.bit_length() - 1, base, {a: round(b, 3) for a, b in st.items()}), flush=True)
    for v in values:
        os.environ[var] = str(v)
        t, st = timed(n)
        print("   %s=%-5s %.3f ms %s" % (var, v, t, {a: round(b, 3) for a, b in st.items()}), flush=True)
    del os.environ[var]


sweep(1 << 21, "ECGPU_MSM_SEG", [1, 2, 4, 8, 16])
sweep(1 << 21, "ECGPU_MSM_CHUNK", [32, 48, 64, 96, 128, 192, 256])
sweep(1 << 21, "ECGPU_MSM_TILE_LOG2", [16, 17, 18])
sweep(1 << 24, "ECGPU_MSM_CHUNK", [256, 384, 512, 768])
sweep(1 << 24, "ECGPU_MSM_SEG", [2, 4, 8])
