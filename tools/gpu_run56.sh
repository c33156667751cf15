#!/bin/bash
# Closing run of the round: default bench line (with frac_note), the curve-by-curve differential over nine parameter sets.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 200 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_fixed_k256_v15.json
timeout 200 bash tools/gpu_differential.sh > gpurun_out/differential_v16.txt 2>&1
tail -12 gpurun_out/differential_v16.txt
