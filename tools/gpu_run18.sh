#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run18
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
for w in var_k256 var_p256; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
echo done
