#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run40
echo "== pytest gpu (msm)"; timeout 1800 python -m pytest tests -m gpu -q -x -k "msm or lincomb" 2>&1 | tail -6 | tee gpurun_out/run40/pytest_gpu.txt
for s2 in 0 1; do
for wl in msm_k256; do
  ECGPU_MSM_SORT2=$s2 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --check 2>/dev/null | tail -1 | tee gpurun_out/run40/bench_${wl}_$s2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sort2=$s2', d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'check', d.get('check_vs_oracle'))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/run40/prof -- python $GRAFT_REPO_ROOT/bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/run40/prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msm' in r['Name']:
        print(r['Name'].split('(')[0][-45:], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
cp "$f" gpurun_out/run40/msm_k256_kernel_stats.csv; rm -rf gpurun_out/run40/prof
