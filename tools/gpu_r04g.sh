#!/bin/bash
# round 4: window width and segment knobs of the 2^21-term k256 MSM re-swept after the sort / tail changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  w=$1; shift
  echo "== $w $*"
  env "$@" python bench.py --only $w --steps 8 --warmup 3 --no-cpu-baseline ${EXTRA:-} 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
}
for c in 14 15 16; do EXTRA="--window $c" run msm_k256_2p21 ECGPU_MSM_GLV=1; done
for c in 13 14 16; do EXTRA="--window $c" run msm_k256_2p21 ECGPU_MSM_GLV=0; done
EXTRA="" run msm_k256_2p21 ECGPU_MSM_SEG=8
EXTRA="" run msm_k256_2p21 ECGPU_MSM_SEG=2
EXTRA="" run msm_k256 ECGPU_MSM_SEG=8
EXTRA="--n 4194304" run msm_k256 ECGPU_MSM_GLV=1
EXTRA="--n 4194304" run msm_k256 ECGPU_MSM_GLV=0
EXTRA="--n 8388608" run msm_k256 ECGPU_MSM_GLV=1
EXTRA="--n 8388608" run msm_k256 ECGPU_MSM_GLV=0
