#!/bin/bash
# round 4: packed two-level sort (k_msm_sort_a / k_msm_sort_b) against the round-3 kernels (ECGPU_MSM_SORT_PACKED=0), the
# to-affine step inside k_msm_combine, k_msm_prepare storing the canonical words it read (k256)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in msm_k256 msm_k256_2p21; do
for pk in 1 0; do
  echo "== $w ECGPU_MSM_SORT_PACKED=$pk"
  ECGPU_MSM_SORT_PACKED=$pk python bench.py --only $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), r.get('stage_ms'))"
done
done
