#!/bin/bash
# round 4: k_msm_accumulate<K256Params> compiled for four waves per SIMD (128 registers + 108 bytes of scratch per lane,
# lib/libecgpu_acc4.so) against the default build (152 registers, three waves per SIMD), alternating on one box
# (the second library: tools/build_alt_lib.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALT=$PWD/elliptic-curves_amd/lib/libecgpu_acc4.so
run() {
  if [ "$3" = "acc4" ]; then export ECGPU_TOOL_LIB=$ALT; else unset ECGPU_TOOL_LIB; fi
  python bench.py --only msm_k256 --n $((1 << $2)) --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $1 $3', round(r['ms_per_step'], 4), r.get('check_vs_oracle'), 'kernel_ms', round(r['roofline']['kernel_ms'], 4), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
}
for v in default acc4 default acc4; do run n24 24 $v; done
for v in default acc4 default acc4; do run n21 21 $v; done
