#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run38
for wl in fixed_k256; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/run38/bench_$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done
