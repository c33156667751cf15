#!/bin/bash
# round 4: the k256 formulas with the differences taken inside the reductions (F::mul_sub / F::sqr_sub)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in fixed_k256 msm_k256 msm_k256_2p21 recover_k256; do
  for rep in 1 2; do
  echo "== $w"
  python bench.py --only $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), r['roofline']['kernel_ms'], {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
  done
done
