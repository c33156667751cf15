#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run10
mkdir -p $OUT
for w in 16 17 18 19 20 21 22; do echo "== bench fixed W=$w"; timeout 600 python bench.py --window $w --steps 10 --warmup 2 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_w$w.json; done
echo "== bench msm default"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
for ch in 256 384; do echo "== bench msm chunk $ch"; ECGPU_MSM_CHUNK=$ch timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256_chunk$ch.json; done
echo done
