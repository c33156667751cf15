#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run8
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
for w in var_p256 var_p384 msm_k256; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
echo done
