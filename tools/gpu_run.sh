#!/bin/bash
# The one GPU-box script.  `gpurun -- 'bash tools/gpu_run.sh <tag> <recipe> [<recipe> ...]'` runs the recipes in order and
# leaves everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/rNN/).
#
# recipes
#   smoke                 __graft_entry__.smoke()
#   pytest[:<expr>]       python -m pytest tests -m gpu [-k <expr>]   (expr with + for spaces: "wycheproof+or+full_size")
#   bench[:<workload>]    python bench.py [--workload W] --check           -> bench_<W>.json
#   bench2                bench.py --gpus 2 as a dry run on one GPU (ranks share device 0, gloo exchange): the N > 1 code path
#   benchN:<N>[:<W>]      the driver's N-rank command as it is, on one GPU: the RCCL canary fails, the run falls back to gloo
#   benchall              the default bench line (all four GPU configs as sub-records) with --check
#   prof:<workload>       rocprofv3 --kernel-trace --stats of bench.py --workload W  -> prof_<W>/ + kernel_stats summary
#   pmc:<workload>        three rocprofv3 --pmc passes (VALU counters, FETCH_SIZE, WRITE_SIZE) of bench.py --workload W
#   env:<VAR=VAL>[,<VAR=VAL>...]:<workload>[:<log2 n>]   the same alternation with environment knobs (ECGPU_MSM_CHUNK=88 ...) instead
#                         of a second library
#   libtest:<suffix>:<expr>   pytest -m gpu -k <expr> (+ for spaces) on lib/libecgpu_<suffix>.so (a build variant under test)
#   san:<asan|tsan>       the host-heavy GPU tests (pipelined host-pointer calls, MSM lanes, asynchronous mode, the group entry points and
#                         its exchange deadline, the table registry) on lib/libecgpu_<kind>.so (make -C elliptic-curves_amd asan tsan):
#                         sanitizer runtime preloaded, reports in san_<kind>.log
#   sweep:<script>        bash tools/<script>.sh
#   ab:<suffix>:<workload>[:<log2 n>]   same-box A/B of lib/libecgpu_<suffix>.so (tools/build_alt_lib.sh) against the default
#                         build, alternating default / alt / default / alt: ms per step, check, dominant kernel, stage times
#   py:<file>             python <file>  (a one-off measurement script under tools/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:?tag}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
ROOT=$PWD
STEPS=${STEPS:-5}        # BENCH_ARGS: extra bench.py arguments for the prof recipe (e.g. "--n 2097152")
for recipe in "$@"; do
  name=${recipe%%:*}; arg=""; [[ "$recipe" == *:* ]] && arg=${recipe#*:}
  echo "=== $recipe"
  case "$name" in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.txt" ;;
    pytest)
      if [ -n "$arg" ]; then timeout 1700 python -m pytest tests -m gpu -q -x --timeout=${TEST_TIMEOUT:-600} -k "${arg//+/ }" --durations=8 > "$OUT/pytest_${arg}.txt" 2>&1; tail -15 "$OUT/pytest_${arg}.txt"
      else timeout 1700 python -m pytest tests -m gpu -q -x --timeout=${TEST_TIMEOUT:-900} --durations=12 > "$OUT/pytest_gpu.txt" 2>&1; tail -20 "$OUT/pytest_gpu.txt"; fi ;;
    bench)
      w=${arg:-fixed_k256}
      timeout 900 python bench.py --workload "$w" --steps "$STEPS" --warmup 2 --check > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; tail -c 3000 "$OUT/bench_$w.json"; tail -3 "$OUT/bench_$w.err" ;;
    benchall)
      timeout 1200 python bench.py --check > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 6000 "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err" ;;
    bench2)   # the N = 2 code path of bench.py on ONE GPU: both ranks on device 0, exchange through the host (gloo)
      ECGPU_BENCH_SHARE_GPU=1 ECGPU_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > "$OUT/bench2.json" 2> "$OUT/bench2.err"
      tail -c 2500 "$OUT/bench2.json"; tail -5 "$OUT/bench2.err" ;;
    benchN)   # N ranks of EXACTLY the driver's command on ONE GPU (ranks share device 0): RCCL cannot form a communicator there, so
              # the canary of sharded.init_exchange fails and the run must go on over the gloo exchange — the first-contact
              # failure the driver's scaling run has to survive.  benchN:2 = the default line, benchN:8:msm_k256 = one workload.
      nn=${arg%%:*}; only=""; [[ "$arg" == *:* ]] && only="--only ${arg#*:}"
      ECGPU_BENCH_SHARE_GPU=1 ECGPU_NCCL_PROBE_TIMEOUT=${PROBE_TIMEOUT:-60} timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$nn" \
        --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus "$nn" --steps 3 --warmup 1 $only > "$OUT/bench_n$nn.json" 2> "$OUT/bench_n$nn.err"
      tail -c 2500 "$OUT/bench_n$nn.json"; tail -5 "$OUT/bench_n$nn.err" ;;
    prof)
      w=${arg:-fixed_k256}
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$w" -o "$w" -- python "$ROOT/bench.py" --only "$w" --steps "$STEPS" --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/prof_$w.log" 2>&1)
      python tools/pmc_summary.py stats "$OUT/prof_$w" | tee "$OUT/kernel_stats_$w.txt"
      find "$OUT/prof_$w" -name "*.db" -delete; find "$OUT/prof_$w" -name "*kernel_trace.csv" -size +1M -delete ;;
    pmc)
      w=${arg:-fixed_k256}
      B="python $ROOT/bench.py --only $w --steps 2 --warmup 1 --no-cpu-baseline"
      i=0
      for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i + 1))
        (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_${w}_$i" -o pmc -- $B > "$OUT/pmc_${w}_$i.log" 2>&1)
      done
      python tools/pmc_summary.py pmc "$OUT" "$w" | tee "$OUT/pmc_summary_$w.txt"
      find "$OUT" -name "*.db" -delete; find "$OUT" -name "*counter_collection.csv" -size +2M -delete ;;
    ab)
      IFS=: read -r suffix w lg <<< "$arg"
      nflag=""; [ -n "${lg:-}" ] && nflag="--n $((1 << lg))"
      for v in default "$suffix" default "$suffix"; do
        if [ "$v" = default ]; then unset ECGPU_TOOL_LIB; else export ECGPU_TOOL_LIB=$ROOT/elliptic-curves_amd/lib/libecgpu_$suffix.so; fi
        timeout 600 python bench.py --only "$w" $nflag --steps "${AB_STEPS:-10}" --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $w ${lg:-} $v', round(r['ms_per_step'], 4), r.get('check_vs_oracle'), 'kernel_ms', round(r['roofline']['kernel_ms'], 4), {k: round(x, 3) for k, x in (r.get('stage_ms') or {}).items()})"
      done 2>&1 | tee -a "$OUT/ab_${suffix}.txt"
      unset ECGPU_TOOL_LIB ;;
    env)
      IFS=: read -r kv w lg <<< "$arg"
      nflag=""; [ -n "${lg:-}" ] && nflag="--n $((1 << lg))"
      for v in default knob default knob; do
        # (the knobs are read by the tool build only, csrc/ecgpu_knobs.h: both legs load it, one with the variables set)
        pre="env ECGPU_TOOL_LIB=$ROOT/elliptic-curves_amd/lib/libecgpu_knobs.so"; [ "$v" = knob ] && pre="$pre ${kv//,/ }"
        timeout 600 $pre python bench.py --only "$w" $nflag --steps "${AB_STEPS:-10}" --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $w ${lg:-} $v $kv', round(r['ms_per_step'], 4), r.get('check_vs_oracle'), 'kernel_ms', round(r['roofline']['kernel_ms'], 4), {k: round(x, 3) for k, x in (r.get('stage_ms') or {}).items()})"
      done 2>&1 | tee -a "$OUT/env_knobs.txt" ;;
    libtest)
      IFS=: read -r suffix expr <<< "$arg"
      ECGPU_TOOL_LIB=$ROOT/elliptic-curves_amd/lib/libecgpu_$suffix.so timeout 1700 python -m pytest tests -m gpu -q -x -k "${expr//+/ }" \
        -p no:cacheprovider > "$OUT/libtest_$suffix.txt" 2>&1; tail -6 "$OUT/libtest_$suffix.txt" ;;
    san)
      RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.$arg-x86_64.so
      # asan: the distribution's runtimes (ROCm's libclang_rt.asan aborts inside its hsa_amd_memory_pool_allocate interceptor here)
      [ "$arg" = asan ] && RT="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
      SEL="pipelined or lanes or asynchronous or group_multi or group_exchange or host_pointer_msm_in_chunks or uniform_schedule_device_resident or comb_table_falls_back"
      LD_PRELOAD=$RT ECGPU_TOOL_LIB=$ROOT/elliptic-curves_amd/lib/libecgpu_$arg.so \
        ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$OUT/san_asan_report UBSAN_OPTIONS=print_stacktrace=1 \
        TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 suppressions=$ROOT/tools/tsan_hip_runtime.supp log_path=$OUT/san_tsan_report" \
        timeout 1500 python -m pytest tests -m gpu -q -k "$SEL" -p no:cacheprovider > "$OUT/san_$arg.log" 2>&1
      tail -8 "$OUT/san_$arg.log"; ls "$OUT" | grep -c "san_${arg}_report" || true
      python tools/san_summary.py "$OUT" "$arg" | tee "$OUT/san_${arg}_summary.txt" ;;
    sweep) timeout 1500 bash "tools/$arg.sh" 2>&1 | tee "$OUT/sweep_$arg.txt" | tail -40 ;;
    py) timeout 1500 python "$arg" 2>&1 | tee "$OUT/py_$(basename "$arg" .py).txt" | tail -60 ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
du -sh "$OUT"
