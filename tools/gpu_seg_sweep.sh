#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for seg in 4 8 16 32 64; do
  for n in 4096 65536 1048576 16777216; do
    ECGPU_MSM_SEG=$seg timeout 600 python bench.py --workload msm_k256 --n $n --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg $seg n', d['config']['units_per_gpu'], 'ms/step %.3f'%d['ms_per_step'])"
  done
done
