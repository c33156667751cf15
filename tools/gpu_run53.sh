#!/bin/bash
# bp384 bring-up: its GPU parity tests, then 2^20 rates.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 280 python -m pytest tests -m gpu -x -q -k "bp384" > gpurun_out/pytest_bp384.log 2>&1
tail -5 gpurun_out/pytest_bp384.log
CID=8 LB=48 timeout 120 bash tools/gpu_run52.sh 2>&1 | tee gpurun_out/bp384_rates.log
