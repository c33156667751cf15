#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests -m gpu -q -x -k "pipelined or pinned" 2>&1 | tail -2
python - <<'PY'
import importlib, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(5)
n = 1 << 24
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
e.mul_by_generator_dev(0, k, n, pts, None)
oxy = torch.empty((1, 64), dtype=torch.uint8, device="cuda"); oinf = torch.empty((16,), dtype=torch.uint8, device="cuda")
def run(name, scal):
    e.lincomb_dev(0, scal, pts, None, n, oxy, oinf)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): e.lincomb_dev(0, scal, pts, None, n, oxy, oinf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("msm k256 2^24 %-28s %.2f ms  (sort %.2f, accumulate %.2f, reduce %.2f)" % (name, dt * 1e3, e.last_timing("sort"), e.last_timing("accumulate"), e.last_timing("reduce")))
run("random scalars", k)
same = k[:1].repeat(n, 1).contiguous()
run("all scalars equal", same)
ones = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); ones[:, 31] = 1
run("all scalars = 1", ones)
half = k.clone(); half[::2] = k[0]
run("every other scalar equal", half)
PY
