#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixed_k256', '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done
python - <<'PY'
import importlib, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0); e.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(3)
for cid, L, name in ((1, 32, "p256"), (2, 48, "p384")):
    k = torch.randint(0, 256, (n, L), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
    out = torch.empty((n, 2 * L), dtype=torch.uint8, device="cuda"); inf = torch.empty((n + 16,), dtype=torch.uint8, device="cuda")
    for _ in range(3): e.mul_by_generator_dev(cid, k, n, out, inf)
    print("%s fixed-base 2^20: kernel %.3f ms, total %.3f ms" % (name, e.last_timing("main"), e.last_timing("total")))
PY
