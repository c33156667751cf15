#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run17
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
for r in 1 2 3; do echo "== bench default"; timeout 900 python bench.py --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_$r.json; done
echo done
