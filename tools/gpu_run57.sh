#!/bin/bash
# After the brainpool t1 twists: the whole GPU suite, their 2^20 rates, a short randomised differential run over eleven sets.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 170 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_v23.txt 2>&1
tail -4 gpurun_out/pytest_gpu_v23.txt
(CID=9 LB=32 timeout 60 bash tools/gpu_run52.sh; CID=10 LB=48 timeout 60 bash tools/gpu_run52.sh) 2>&1 | grep curve | tee gpurun_out/bpt1_rates.txt
FUZZ_SECONDS=45 FUZZ_SEED=1111 timeout 100 bash tools/gpu_fuzz.sh > gpurun_out/fuzz_v18.txt 2>&1
tail -3 gpurun_out/fuzz_v18.txt
