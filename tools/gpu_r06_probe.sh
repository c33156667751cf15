#!/bin/bash
# round 6: does the order of the workloads in one bench process decide what a 2^21-term share costs?  The same workload alone in a
# fresh process, and after a heavy one in the same process (sweep: recipe of tools/gpu_run.sh).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
python bench.py --only msm_k256 --n 2097152 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('alone: ms_per_step', round(r['ms_per_step'], 4), r['check_vs_oracle'], {k: round(v, 3) for k, v in r['stage_ms'].items()})"
done
python - <<'P'
import importlib, json, sys, types
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
import bench
args = types.SimpleNamespace(gpus=1, steps=20, warmup=5, only=None, n=0, window=0, sets=4, check=False, no_check=False, sync_calls=False,
                             no_cpu_baseline=True, no_extras=True, cpu_plumbing=False, group_msm_child=0)
b = bench.Bench(args)
for name in ("msm_k256_2p21", "var_p384", "msm_k256_2p21", "msm_k256", "msm_k256_2p21", "msm_k256_2p21_lanes", "msm_k256_2p21_sharded_lanes", "fixed_k256_ct", "lincomb_ct_k256"):
    r = b.run(name, False)
    print(name, "ms_per_step", round(r["ms_per_step"], 4), r["check_vs_oracle"], "kernel_ms", r["roofline"]["kernel_ms"], {k: round(v, 3) for k, v in r["stage_ms"].items()}, flush=True)
b.close()
P
