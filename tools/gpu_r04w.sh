#!/bin/bash
# round 4: Field::mul_sub / sqr_sub on the Montgomery fields (the subtrahend in the high columns) against the same source built
# with -DECGPU_FUSED_SUB=0 for the two ladders (lib/libecgpu_nofused.so), alternating on one box
# (the second library: tools/build_alt_lib.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALT=$PWD/elliptic-curves_amd/lib/libecgpu_nofused.so
run() {
  echo "== $1 $2"
  if [ "$2" = "nofused" ]; then export ECGPU_TOOL_LIB=$ALT; else unset ECGPU_TOOL_LIB; fi
  python bench.py --only $1 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['ms_per_step'], r.get('check_vs_oracle'), 'kernel_ms', round(r['roofline']['kernel_ms'], 4), 'min', round(r['roofline']['kernel_ms_min'], 4))"
}
for w in var_p256 var_p384; do
  for v in fused nofused fused nofused; do run $w $v; done
done
