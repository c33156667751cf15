#!/bin/bash
# round 4: k_msm_bucket_finish<K256Params> back at two waves per SIMD (launch bounds; 16 bytes of scratch): per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ROOT=$PWD
for lg in 21 24; do
  out=/tmp/r04v2_$lg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $ROOT/bench.py --only msm_k256 --n $((1 << lg)) --steps 10 --warmup 3 --no-cpu-baseline > $out.log 2>&1)
  echo "== n=2^$lg"; grep -o '"ms_per_step": *[0-9.]*\|"check_vs_oracle": *[a-z]*' $out.log | tr '\n' ' '; echo
  python tools/pmc_summary.py stats $out | grep -E "k_msm_(bucket_finish|reduce_segments|accumulate|combine|prepare)" | sed 's/^/     /'
done
