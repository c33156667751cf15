#!/bin/bash
# MSM window sweep at several sizes (input to msm_window_bits in ecgpu_msm.h)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import importlib, time, numpy as np, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0); e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(11)
nmax = 1 << 22
k = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
pts = torch.empty((nmax, 64), dtype=torch.uint8, device="cuda")
e.mul_by_generator_dev(0, k, nmax, pts, None)
r = torch.empty((1, 64), dtype=torch.uint8, device="cuda"); ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
for lg in (12, 14, 16, 17, 18, 19, 20, 21, 22):
    n = 1 << lg
    row = []
    for c in range(max(4, lg - 10), min(16, lg - 3) + 1):
        e.set_msm_window(c)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); e.lincomb_dev(0, k[:n], pts[:n], None, n, r, ri); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append((c, min(ts[1:]) * 1e3))
    best = min(row, key=lambda t: t[1])
    print("n=2^%d auto c=%d | " % (lg, min(16, max(4, lg - 7))) + " ".join("c%d:%.2f" % t for t in row) + " | best c=%d %.2f ms" % best)
PY
