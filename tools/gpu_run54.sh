#!/bin/bash
# After bp384: the whole GPU suite, a randomised differential run over nine parameter sets, the curve-by-curve differential.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_v22.txt 2>&1
tail -4 gpurun_out/pytest_gpu_v22.txt
FUZZ_SECONDS=100 FUZZ_SEED=384 timeout 170 bash tools/gpu_fuzz.sh > gpurun_out/fuzz_v17.txt 2>&1
tail -6 gpurun_out/fuzz_v17.txt
