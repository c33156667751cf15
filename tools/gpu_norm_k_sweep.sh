#!/bin/bash
# k_normalize with the quad-major hand-over: points per inversion (ECGPU_NORM_K, tool build) at 2^20 k256 points — was the round-5
# optimum (16 = one wave per SIMD) moved by the cheaper per-point passes?   (sweep: recipe of tools/gpu_run.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ECGPU_TOOL_LIB=$PWD/elliptic-curves_amd/lib/libecgpu_knobs.so
for k in 16 8 10 12 14 16 20 24 32 16; do
  ECGPU_NORM_K=$k python bench.py --only fixed_k256 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NORM_K=$k ms_per_step', round(r['ms_per_step'], 4), r['check_vs_oracle'], {k: round(v, 4) for k, v in r['stage_ms'].items()})"
done
