#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run36
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/run36/pytest_gpu.txt
python - <<'PY'
import importlib, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
n = 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(3)
pk = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
msg = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
sig = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
ok = torch.empty((n,), dtype=torch.uint8, device="cuda")
for _ in range(2): e.schnorr_verify_raw_dev(pk, msg, 32, sig, n, ok)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): e.schnorr_verify_raw_dev(pk, msg, 32, sig, n, ok)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("bip340 verify_raw 2^20 (random keys, ~half lift): %.3f ms -> %.3e verifications/s; ok sum %d" % (dt * 1e3, n / dt, int(ok.sum())))
PY
