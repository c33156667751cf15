#!/bin/bash
# round 4: the Horner chain of the MSM on quad lanes (complete doublings / additions in homogeneous coordinates) and the
# level-A flush of k_msm_prepare (workgroups per launch, rotated flush order): per-kernel times under rocprofv3 --kernel-trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ROOT=$PWD
prof() {   # prof <label> <log2 n> [ENV=..]...
  local label=$1 lg=$2; shift 2
  local out=/tmp/r04t_$label
  (cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $ROOT/bench.py --only msm_k256 --n $((1 << lg)) --steps 10 --warmup 3 --no-cpu-baseline > $out.log 2>&1)
  python - "$out.log" "$label" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if l:
    r = json.loads(l[-1])
    print("== %s: %.3f ms/step under the profiler, check=%s, stages=%s" % (sys.argv[2], r["ms_per_step"], r.get("check_vs_oracle"), {k: round(v, 3) for k, v in r["stage_ms"].items()}))
else:
    print("== %s FAILED" % sys.argv[2])
PY
  python tools/pmc_summary.py stats $out | grep -E "k_msm_(prepare|combine|sort_a|sort_b |accumulate|bucket_finish|reduce_segments|scan)" | sed 's/^/     /'
}
prof n19 19
prof n20 20
prof n21 21
prof n21_reps2 21 ECGPU_MSM_PREP_REPS=2
prof n21_reps4 21 ECGPU_MSM_PREP_REPS=4
prof n21_reps4_rot 21 ECGPU_MSM_PREP_REPS=4 ECGPU_MSM_PREP_ROT=1
prof n21_rot 21 ECGPU_MSM_PREP_ROT=1
prof n24 24
prof n24_rot 24 ECGPU_MSM_PREP_ROT=1
prof n24_reps16 24 ECGPU_MSM_PREP_REPS=16
echo "== without the profiler"
for lg in 21 24; do
  python bench.py --only msm_k256 --n $((1 << lg)) --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=2^$lg', r['ms_per_step'], r.get('check_vs_oracle'), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
done
