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
#   sweep:<script>        bash tools/<script>.sh
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
      if [ -n "$arg" ]; then timeout 1700 python -m pytest tests -m gpu -q -x -k "${arg//+/ }" --durations=8 > "$OUT/pytest_${arg}.txt" 2>&1; tail -15 "$OUT/pytest_${arg}.txt"
      else timeout 1700 python -m pytest tests -m gpu -q -x --durations=12 > "$OUT/pytest_gpu.txt" 2>&1; tail -20 "$OUT/pytest_gpu.txt"; fi ;;
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
    sweep) timeout 1500 bash "tools/$arg.sh" 2>&1 | tee "$OUT/sweep_$arg.txt" | tail -40 ;;
    py) timeout 1500 python "$arg" 2>&1 | tee "$OUT/py_$(basename "$arg" .py).txt" | tail -60 ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
du -sh "$OUT"
