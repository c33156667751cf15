#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run26
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== bench msm"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
echo "== skewed"; python - <<'PY'
import importlib, time, sys, numpy as np, torch
sys.path.insert(0, "tests")
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
n = 1 << 24
g = torch.Generator(device="cuda"); g.manual_seed(5)
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
e.mul_by_generator_dev(0, k, n, pts, None)
r = torch.empty((1, 64), dtype=torch.uint8, device="cuda"); ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
ones = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); ones[:, 31] = 1
same = k[:1].repeat(n, 1).contiguous()
for name, s in (("random", k), ("all ones", ones), ("all equal", same)):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time(); e.lincomb_dev(0, s, pts, None, n, r, ri); torch.cuda.synchronize(); dt = time.time() - t0
    print("msm 2^24 %-10s %.1f ms" % (name, dt * 1e3))
PY
echo done
