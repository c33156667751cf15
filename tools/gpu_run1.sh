#!/bin/bash
# First GPU pass: smoke, instruction-rate probes, parity tests, bench lines, rocprof summary.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run1
mkdir -p $OUT
echo "== rocminfo" ; rocminfo | grep -E "Marketing Name|gfx9|Compute Unit|Max Clock" | head -8
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== probes"
timeout 300 python - <<'PY' 2>&1 | tee $OUT/probes.txt
import importlib
ecgpu = importlib.import_module("elliptic-curves_amd")
e = ecgpu.Engine(0)
names = ["v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32+add", "v_add_u32", "add_u64", "v_mad_u32_u24", "v_fma_f64", "add64+shift"]
for i, nm in enumerate(names):
    print("%-18s %.3e ops/s" % (nm, e.valu_probe(i)))
PY
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check 2>&1 | tail -3 | tee $OUT/bench_fixed_k256.json
for w in 4 8 12; do echo "== bench fixed window $w"; timeout 600 python bench.py --steps 5 --warmup 1 --window $w --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_w$w.json; done
echo "== bench msm 2^20"; timeout 600 python bench.py --workload msm_k256 --n 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_2p20.json
echo "== bench msm 2^24"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
echo "== bench var p256"; timeout 900 python bench.py --workload var_p256 --steps 2 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_var_p256.json
echo "== bench var p384"; timeout 900 python bench.py --workload var_p384 --steps 1 --warmup 1 --n 262144 --check 2>&1 | tail -1 | tee $OUT/bench_var_p384.json
echo "== rocprof fixed"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_fixed -o fixed -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_fixed.log 2>&1)
find $OUT/prof_fixed -name "*stats*" | head; for f in $(find $OUT/prof_fixed -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
echo "== rocprof msm"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_msm -o msm -- python $OLDPWD/bench.py --workload msm_k256 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_msm.log 2>&1)
for f in $(find $OUT/prof_msm -name "*kernel_stats*.csv" | head -1); do head -14 $f; done
# keep only the small summaries
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace*" -size +2M -delete
echo done
