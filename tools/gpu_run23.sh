#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run23
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== timings"; python - <<'PY'
import importlib, time, numpy as np, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
n = 1 << 20
for cid, L, name in ((0, 32, "k256"), (1, 32, "p256"), (2, 48, "p384")):
    xs = torch.randint(0, 256, (n, L), dtype=torch.uint8, device="cuda")
    xs[:, 0] = 0x7f
    odd = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda")
    out = torch.empty((n, 2 * L), dtype=torch.uint8, device="cuda"); ok = torch.empty((n + 16,), dtype=torch.uint8, device="cuda")
    dp = lambda t: t.data_ptr()
    for _ in range(2):
        e._chk(e._lib.ecgpu_batch_decompress_dev(e._ctx, cid, ec._dp(xs), ec._dp(odd), n, ec._dp(out), ec._dp(ok)))
    print("%s decompress 2^20: %.3f ms (%.3g /s), ok fraction %.3f" % (name, e.last_timing("main"), n / (e.last_timing("main") * 1e-3), float(ok[:n].float().mean())))
PY
echo done
