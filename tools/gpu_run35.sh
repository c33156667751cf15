#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run35
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/run35/pytest_gpu.txt
python - <<'PY'
import importlib, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
n = 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
out = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); inf = torch.empty((n + 16,), dtype=torch.uint8, device="cuda")
for _ in range(3): e.mul_by_generator_dev(3, k, n, out, inf)
print("sm2 fixed-base 2^20: kernel %.3f ms, total %.3f ms" % (e.last_timing("main"), e.last_timing("total")))
pts = out.clone()
for _ in range(2): e.mul_dev(3, k, pts, None, n, out, inf)
print("sm2 variable-base 2^20: kernel %.3f ms" % e.last_timing("main"))
PY
