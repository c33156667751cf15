#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run13
mkdir -p $OUT
for k in 4 6 8 10 12 16 24 32; do echo "== bench fixed NORM_K=$k"; ECGPU_NORM_K=$k timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_k$k.json; done
echo done
