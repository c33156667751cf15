#!/bin/bash
# round 4: level-B workgroup size (ECGPU_MSM_SORTB_T) and accumulation chunk (ECGPU_MSM_CHUNK) sweeps at 2^24 / 2^21 terms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {   # workload, env assignments...
  w=$1; shift
  echo "== $w $*"
  env "$@" python bench.py --only $w --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
}
for w in msm_k256 msm_k256_2p21; do
  for t in 1024 512 256; do run $w ECGPU_MSM_SORTB_T=$t; done
done
for c in 232 348 696 928; do run msm_k256 ECGPU_MSM_CHUNK=$c; done
for c in 32 48 96 128; do run msm_k256_2p21 ECGPU_MSM_CHUNK=$c; done
