#!/bin/bash
# round 4: k_msm_prepare with the streamed digits; the tail kernels of the 2^21-term MSM under the counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ROOT=$PWD
prof() {   # prof <label> <log2 n> [ENV=..]...
  local label=$1 lg=$2; shift 2
  local out=/tmp/r04u_$label
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
  python tools/pmc_summary.py stats $out | grep -E "k_msm_" | sed 's/^/     /'
}
prof n21 21
prof n24 24
if [ -n "${PMC:-}" ]; then
echo "== counters, 2^21 terms"
OUT=/tmp/r04u_pmc
mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_msm_k256_1 -o pmc -- python $ROOT/bench.py --only msm_k256 --n 2097152 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_msm_k256_2 -o pmc -- python $ROOT/bench.py --only msm_k256 --n 2097152 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1)
python tools/pmc_summary.py pmc $OUT msm_k256 | grep -E "bucket_finish|reduce_segments|reduce_windows|window_sums|k_msm_combine|k_msm_prepare"
fi
echo "== without the profiler"
for lg in 21 24; do
  python bench.py --only msm_k256 --n $((1 << lg)) --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=2^$lg', r['ms_per_step'], r.get('check_vs_oracle'), {k: round(v, 3) for k, v in r.get('stage_ms').items()})"
done
