#!/bin/bash
# round 4, first contact: k_normalize_wg (one inversion per workgroup) against k_normalize, and a forecast for running the
# two term-halves of one MSM on two lanes (independent MSMs of half the size with 1 / 2 lanes in flight)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for wg in 1 0; do
  echo "== fixed_k256 ECGPU_NORM_WG=$wg"
  ECGPU_NORM_WG=$wg python bench.py --only fixed_k256 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), r.get('stage_ms'))"
done
for w in var_p256 var_p384; do
for wg in 1 0; do
  echo "== $w ECGPU_NORM_WG=$wg"
  ECGPU_NORM_WG=$wg python bench.py --only $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), r.get('stage_ms'))"
done
done
echo "== lanes forecast"
python tools/gpu_msm_lanes.py 20 23
