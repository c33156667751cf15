#!/usr/bin/env python3
"""GLV halves against the plain scalar, and the window width, around one GPU's share of the sharded 2^24-term k256 MSM
(the last crossover check predates the Horner chain on the rows of a wave).  For every size, GLV mode
and window width: the step time (best of `reps` synchronous calls, wall clock around a drained stream) and the stage events.
Runs on the tool build of the library (lib/libecgpu_knobs.so: ECGPU_MSM_GLV is read there only).
    python tools/gpu_msm_crossover.py [log2 sizes ...]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ec = importlib.import_module("elliptic-curves_amd")
sizes = [int(a) for a in sys.argv[1:]] or [20, 21, 22]
e = ec.Engine(0, variant="knobs")   # the tool build: the ECGPU_* knobs below are read there only (csrc/ecgpu_knobs.h)
e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda")
g.manual_seed(13)
nmax = 1 << max(sizes)
NSETS = 3
sets = []
for j in range(NSETS):
    s = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g)
    s[:, 0] &= 0x7F
    pts = torch.empty((nmax, 64), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    e.mul_by_generator_dev(0, s, nmax, pts, None)
    k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g)
    k[:, 0] &= 0x7F
    sets.append((k, pts))
r = torch.empty((1, 64), dtype=torch.uint8, device="cuda")
ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
STAGES = ("prepare", "sort", "accumulate", "finish", "tree", "combine")


def timed(n, c, reps=9):
    e.set_msm_window(c)
    ts, st = [], {}
    for i in range(reps):
        k, pts = sets[i % NSETS]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.lincomb_dev(0, k[:n], pts[:n], None, n, r, ri)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if i:
            for s in STAGES:
                v = e.last_timing(s)
                if v is not None:
                    st.setdefault(s, []).append(v)
    ts = sorted(ts[1:])
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3, {s: round(min(v), 3) for s, v in st.items()}


for lg in sizes:
    n = 1 << lg
    for glv in ("0", "1"):
        os.environ["ECGPU_MSM_GLV"] = glv
        for c in range(12, 17):
            best, med, st = timed(n, c)
            print("n=2^%d glv=%s c=%2d  best %.3f  median %.3f ms  %s" % (lg, glv, c, best, med, st), flush=True)
    del os.environ["ECGPU_MSM_GLV"]
    best, med, st = timed(n, 0)
    print("n=2^%d auto        best %.3f  median %.3f ms  %s" % (lg, best, med, st), flush=True)
