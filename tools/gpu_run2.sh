#!/bin/bash
# Second GPU pass: exact instruction-rate probes, full parity tests, MSM after the top-window fix,
# rocprofv3 kernel stats (csv) and PMC counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run2
mkdir -p $OUT
echo "== isa probes"
timeout 300 python - <<'PY' 2>&1 | tee $OUT/isa_probes.txt
import importlib
ecgpu = importlib.import_module("elliptic-curves_amd")
e = ecgpu.Engine(0)
names = ["v_mad_u64_u32", "v_add_u32", "v_add_co+v_addc_co (per instr)", "v_mov_b32", "v_lshl_add_u64", "v_mul_lo_u32", "v_mul_hi_u32",
         "v_and_b32", "v_mad_u32_u24", "v_alignbit_b32", "v_add3_u32", "v_lshlrev_b64", "v_fma_f64", "v_mul_u32_u24", "v_lshl_or_b32", "v_cndmask_b32"]
for i, nm in enumerate(names):
    v = e.valu_probe(100 + i)
    print("%-34s %.3e lane-ops/s = %.3e wave-instr/s -> %.2f cycles/wave-instr/SIMD @2.4GHz" % (nm, v, v / 64, 1024 * 2.4e9 / (v / 64)))
PY
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== bench msm 2^24"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
for c in 13 14 15; do echo "== bench msm 2^24 c=$c"; timeout 600 python bench.py --workload msm_k256 --steps 2 --warmup 1 --window $c --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256_c$c.json; done
echo "== bench msm 2^20"; timeout 600 python bench.py --workload msm_k256 --n 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_2p20.json
echo "== rocprof stats fixed"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_fixed -o fixed -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_fixed.log 2>&1)
find $OUT/prof_fixed -name "*.csv" | head; for f in $(find $OUT/prof_fixed -name "*kernel_stats.csv" | head -1); do cat $f | head -12; done
echo "== rocprof stats msm"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_msm -o msm -- python $OLDPWD/bench.py --workload msm_k256 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_msm.log 2>&1)
for f in $(find $OUT/prof_msm -name "*kernel_stats.csv" | head -1); do cat $f | head -16; done
echo "== pmc valu fixed"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/pmc_valu_fixed -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/pmc_valu_fixed.log 2>&1)
for f in $(find $OUT/pmc_valu_fixed -name "*counter_collection.csv" | head -1); do grep -E "Counter_Name|k_fixed_base" $f | head -20; done
echo "== pmc fetch fixed"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch_fixed -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/pmc_fetch_fixed.log 2>&1)
for f in $(find $OUT/pmc_fetch_fixed -name "*counter_collection.csv" | head -1); do grep -E "Counter_Name|k_fixed_base" $f | head -8; done
echo "== pmc write fixed"
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write_fixed -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/pmc_write_fixed.log 2>&1)
for f in $(find $OUT/pmc_write_fixed -name "*counter_collection.csv" | head -1); do grep -E "Counter_Name|k_fixed_base" $f | head -8; done
# trim: keep csv summaries only, drop big traces
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
du -sh $OUT
echo done
