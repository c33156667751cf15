#!/bin/bash
# round 4: accumulation chunk at 2^21 terms (two instead of three rounds of lanes: fewer partial sums per bucket for the finish)
# and the GLV threshold at 2^22 terms after the cheaper split
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {   # run <label> <log2 n> [ENV=..]...
  local label=$1 lg=$2; shift 2
  env "$@" python bench.py --only msm_k256 --n $((1 << lg)) --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $label', round(r['ms_per_step'], 4), r.get('check_vs_oracle'), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
}
for rep in 1 2; do
  run n21_default 21
  run n21_chunk96 21 ECGPU_MSM_CHUNK=96
  run n21_chunk128 21 ECGPU_MSM_CHUNK=128
  run n21_chunk192 21 ECGPU_MSM_CHUNK=192
done
run n22_default 22
run n22_glv 22 ECGPU_MSM_GLV_MAX_LOG2=22
run n22_default 22
run n22_glv 22 ECGPU_MSM_GLV_MAX_LOG2=22
run n20_default 20
run n20_chunk64 20 ECGPU_MSM_CHUNK=64
run n20_chunk96 20 ECGPU_MSM_CHUNK=96
