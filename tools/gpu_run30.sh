#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for n in 4096 65536 1048576 16777216; do
  timeout 600 python bench.py --workload msm_k256 --n $n --steps 5 --warmup 2 --no-cpu-baseline $( [ $n -le 65536 ] && echo --check ) 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['config']['units_per_gpu'], 'ms/step %.3f'%d['ms_per_step'], d.get('check_vs_oracle'))"
done
