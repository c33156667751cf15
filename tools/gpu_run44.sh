#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run44
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/run44/pytest_gpu.txt
for wl in var_k256; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --check 2>/dev/null | tail -1 | tee gpurun_out/run44/bench_$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done
python - <<'PY'
import importlib, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
e.set_stream(torch.cuda.current_stream().cuda_stream)
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
print("bip340 verify_raw 2^20 (random keys, ~half lift): %.3f ms -> %.3e verifications/s" % (dt * 1e3, n / dt))
PY
