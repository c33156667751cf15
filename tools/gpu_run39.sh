#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run39
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/run39/pytest_gpu.txt
for wl in fixed_k256 msm_k256 msm_p256; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | tee gpurun_out/run39/bench_$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done
