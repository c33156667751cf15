#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run48
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/run48/pytest_gpu.txt
python - <<'PY' | tee gpurun_out/run48/host_msm.txt
import importlib, time, numpy as np
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
rng = np.random.default_rng(2)
for lg in (22, 23, 24):
    n = 1 << lg
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8); k[:, 0] &= 0x7f
    pts, _ = e.mul_by_generator(0, k)
    k2 = rng.integers(0, 256, (n, 32), dtype=np.uint8); k2[:, 0] &= 0x7f
    e.lincomb(0, k2.reshape(-1), pts)
    t = time.perf_counter(); reps = 3
    for _ in range(reps): o, f = e.lincomb(0, k2.reshape(-1), pts)
    dt = (time.perf_counter() - t) / reps
    print("host-pointer msm k256 2^%d (pageable buffers): %.2f ms -> %.3e terms/s" % (lg, dt * 1e3, n / dt))
PY
