#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for wl in var_p256 var_k256 ecdsa_p256; do
  python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --check 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done
