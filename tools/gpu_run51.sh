#!/bin/bash
# small MSM: bucket method vs per-term multiplication + tree sum
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import importlib, os, time, torch
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0); e.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(11)
for cid, L, name in ((0, 32, "k256"), (1, 32, "p256"), (2, 48, "p384")):
    nmax = 1 << 19
    k = torch.randint(0, 256, (nmax, L), dtype=torch.uint8, device="cuda", generator=g); k[:, 0] &= 0x7f
    pts = torch.empty((nmax, 2 * L), dtype=torch.uint8, device="cuda")
    e.mul_by_generator_dev(cid, k, nmax, pts, None)
    k2 = torch.randint(0, 256, (nmax, L), dtype=torch.uint8, device="cuda", generator=g); k2[:, 0] &= 0x7f
    r = torch.empty((1, 2 * L), dtype=torch.uint8, device="cuda"); ri = torch.empty((16,), dtype=torch.uint8, device="cuda")
    for lg in (4, 8, 10, 12, 14, 15, 16, 17, 18, 19):
        n = 1 << lg
        res = {}
        for mode in ("-1", "24"):
            os.environ["ECGPU_MSM_SMALL_LOG2"] = mode
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); e.lincomb_dev(cid, k2[:n], pts[:n], None, n, r, ri); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[mode] = min(ts[1:]) * 1e3; res["o" + mode] = bytes(r.cpu().numpy())
        assert res["o-1"] == res["o24"], (name, lg)
        print("%s n=2^%d: buckets %.3f ms, per-term + tree %.3f ms" % (name, lg, res["-1"], res["24"]))
PY
