#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run20
mkdir -p $OUT
for p in 0 1 0 1; do echo "== bench fixed prefetch=$p"; ECGPU_FIXED_PREFETCH=$p timeout 600 python bench.py --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_p$p.json; done
for p in 0 1; do echo "== bench fixed W=20 prefetch=$p"; ECGPU_FIXED_PREFETCH=$p timeout 600 python bench.py --window 20 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1; done
echo done
