#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 26 27 29; do
  timeout 600 python bench.py --window $w --steps 20 --warmup 3 --no-cpu-baseline --check 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W=$w', '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))" || echo "W=$w failed"
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
