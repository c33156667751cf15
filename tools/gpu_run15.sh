#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run15
mkdir -p $OUT
for w in 16 18 19 20 22; do echo "== bench fixed W=$w"; timeout 600 python bench.py --window $w --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_w$w.json; done
for n in 262144 2097152 8388608; do echo "== bench fixed n=$n"; timeout 600 python bench.py --n $n --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_n$n.json; done
echo done
