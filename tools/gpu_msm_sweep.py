#!/usr/bin/env python3
"""MSM window sweep on the GPU box (input to msm_window_bits in ecgpu_msm.h) + stage times at the automatic width.
    python tools/gpu_msm_sweep.py [curve] [max_log2]        (run through tools/gpu_run.sh ... py:tools/gpu_msm_sweep.py)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ec = importlib.import_module("elliptic-curves_amd")
curve = sys.argv[1] if len(sys.argv) > 1 else "k256"
maxlg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cid = ec.CURVE_IDS[curve]
L = ec.FIELD_BYTES[cid]
e = ec.Engine(0, variant="knobs")   # the tool build: the ECGPU_* knobs below are read there only (csrc/ecgpu_knobs.h)
e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda")
g.manual_seed(11)
nmax = 1 << maxlg
k = torch.randint(0, 256, (nmax, L), dtype=torch.uint8, device="cuda", generator=g)
k[:, 0] &= 0x7F
pts = torch.empty((nmax, 2 * L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
e.mul_by_generator_dev(cid, k, nmax, pts, None)
k = torch.randint(0, 256, (nmax, L), dtype=torch.uint8, device="cuda", generator=g)
k[:, 0] &= 0x7F
r = torch.empty((1, 2 * L), dtype=torch.uint8, device="cuda")
ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def timed(n, c, reps=4):
    e.set_msm_window(c)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.lincomb_dev(cid, k[:n], pts[:n], None, n, r, ri)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3


modes = ("0", "1") if curve == "k256" else ("0",)
for lg in range(14, maxlg + 1):
    n = 1 << lg
    if lg < 16 and lg % 2:
        continue
    for glv in modes:
        os.environ["ECGPU_MSM_GLV"] = glv
        row = [(c, timed(n, c)) for c in range(max(9, lg - 9), 17)]
        best = min(row, key=lambda t: t[1])
        print("n=2^%d glv=%s | " % (lg, glv) + " ".join("c%d:%.2f" % t for t in row) + " | best c=%d %.3f ms" % best, flush=True)
    del os.environ["ECGPU_MSM_GLV"]
    auto = timed(n, 0)
    st = {s: e.last_timing(s) for s in ("sort", "accumulate", "reduce", "normalize", "total")}
    print("n=2^%d auto %.3f ms  %.3g terms/s  stages %s"
          % (lg, auto, n / auto * 1e3, {a: round(b, 3) for a, b in st.items() if b is not None}), flush=True)
