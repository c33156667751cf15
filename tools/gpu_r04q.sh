#!/bin/bash
# round 4: k256 formulas with the differences taken inside the reductions (F::mul_sub / F::sqr_sub; lib/libecgpu.so) against the
# same library built with -DECGPU_FUSED_SUB=0 (lib/libecgpu_nofused.so), alternating on one box
# (the second library: tools/build_alt_lib.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALT=$PWD/elliptic-curves_amd/lib/libecgpu_nofused.so
run() {
  echo "== $1 $2"
  if [ "$2" = "nofused" ]; then export ECGPU_TOOL_LIB=$ALT; else unset ECGPU_TOOL_LIB; fi
  python bench.py --only $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), 'kernel_ms', round(r['roofline']['kernel_ms'], 4), 'min', round(r['roofline']['kernel_ms_min'], 4), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
}
for w in fixed_k256 msm_k256 msm_k256_2p21 recover_k256; do
  for v in fused nofused fused nofused; do run $w $v; done
done
