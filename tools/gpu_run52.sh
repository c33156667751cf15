#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import importlib, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0); e.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(3)
cid, L = int(__import__("os").environ.get("CID", "4")), int(__import__("os").environ.get("LB", "28"))
k = torch.randint(0, 256, (n, L), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= (1 if L == 66 else 0x7f)
out = torch.empty((n, 2 * L), dtype=torch.uint8, device="cuda"); inf = torch.empty((n + 16,), dtype=torch.uint8, device="cuda")
for _ in range(3): e.mul_by_generator_dev(cid, k, n, out, inf)
print("curve %d " % cid + "fixed-base 2^20: kernel %.3f ms, total %.3f ms -> %.3e /s" % (e.last_timing("main"), e.last_timing("total"), n / e.last_timing("total") * 1e3))
pts = out.clone()
for _ in range(2): e.mul_dev(cid, k, pts, None, n, out, inf)
print("curve %d " % cid + "variable-base 2^20: kernel %.3f ms -> %.3e /s" % (e.last_timing("main"), n / e.last_timing("total") * 1e3))
oxy = torch.empty((1, 2 * L), dtype=torch.uint8, device="cuda"); oinf = torch.empty((16,), dtype=torch.uint8, device="cuda")
for _ in range(2): e.lincomb_dev(cid, k, pts, None, n, oxy, oinf)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): e.lincomb_dev(cid, k, pts, None, n, oxy, oinf)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print("curve %d " % cid + "MSM 2^20: %.2f ms -> %.3e terms/s" % (dt * 1e3, n / dt))
PY
