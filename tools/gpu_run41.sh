#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run41
python - <<'PY'
import importlib, time, os, numpy as np, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
rng = np.random.default_rng(1)
for lg in (20, 22):
    n = 1 << lg
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8); k[:, 0] &= 0x7f
    e.mul_by_generator(0, k)
    t = time.perf_counter(); reps = 3
    for _ in range(reps): out, inf = e.mul_by_generator(0, k)
    dt = (time.perf_counter() - t) / reps
    print("host-pointer fixed-base k256 2^%d: %.2f ms -> %.3e /s" % (lg, dt * 1e3, n / dt))
# MSM: sort mode x size
g = torch.Generator(device="cuda"); g.manual_seed(5)
for lg in (16, 18, 19, 20, 21, 22):
    n = 1 << lg
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
    pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); inf = torch.empty((n + 16,), dtype=torch.uint8, device="cuda")
    e.mul_by_generator_dev(0, k, n, pts, inf)
    k2 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); k2[:, 0] &= 0x7f
    oxy = torch.empty((1, 64), dtype=torch.uint8, device="cuda"); oinf = torch.empty((16,), dtype=torch.uint8, device="cuda")
    res = {}
    for s2 in ("0", "1"):
        os.environ["ECGPU_MSM_SORT2"] = s2
        for _ in range(2): e.lincomb_dev(0, k2, pts, None, n, oxy, oinf)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): e.lincomb_dev(0, k2, pts, None, n, oxy, oinf)
        torch.cuda.synchronize(); res[s2] = (time.perf_counter() - t) / 5; res["xy" + s2] = bytes(oxy.cpu().numpy())
    assert res["xy0"] == res["xy1"]
    print("msm k256 2^%d: single-level %.3f ms, two-level %.3f ms" % (lg, res["0"] * 1e3, res["1"] * 1e3))
del os.environ["ECGPU_MSM_SORT2"]
PY
for wl in msm_p256; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g' % d['value'], d['unit'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
