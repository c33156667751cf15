#!/bin/bash
# Randomised differential run: random sizes, window widths, chunk sizes, sort modes, duplicated / cancelling / identity
# terms; every result against the oracle (n <= 2^13) or a property (larger n).  One-off, not part of the test suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import importlib, os, random, sys, time, numpy as np
sys.path.insert(0, "tests")
ec = importlib.import_module("elliptic-curves_amd")
import oracle_lib, pyec
from gpu_common import rand_scalars, scalars_to_int_sum
oracle_lib.build()
e = ec.Engine(0)
rng = random.Random(int(os.environ.get("FUZZ_SEED", "20260924")))
budget = float(os.environ.get("FUZZ_SECONDS", "150"))
t_end = time.time() + budget
stats = {"msm_oracle": 0, "msm_property": 0, "fixed": 0, "var": 0}
pools = {}
def pool(c):
    if c.name not in pools:
        n = 1 << 16
        pts, _ = e.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xF0 + c.cid))
        pools[c.name] = pts.reshape(n, 2 * c.L).copy()
    return pools[c.name]
while time.time() < t_end:
    c = pyec.CURVES[rng.choice(["k256", "p256", "p384", "sm2", "p224", "p192", "p521", "bp256", "bp384", "bp256t1", "bp384t1"])]
    L = c.L
    kind = rng.choice(["msm", "msm", "msm", "fixed", "var"])
    if kind == "msm":
        big = rng.random() < 0.25
        n = rng.randrange(1 << 14, 1 << 19) if big else rng.randrange(1, 1 << 13)
        cb = rng.choice([0, 0, rng.randrange(4, 17)])
        os.environ["ECGPU_MSM_SORT2"] = rng.choice(["0", "1"])
        if rng.random() < 0.3: os.environ["ECGPU_MSM_CHUNK"] = str(rng.choice([1, 2, 3, 7, 33, 500, 100000]))
        else: os.environ.pop("ECGPU_MSM_CHUNK", None)
        os.environ["ECGPU_MSM_SMALL_LOG2"] = rng.choice(["-1", "16"])
        e.set_msm_window(cb)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30)).copy().reshape(n, L)
        if big:
            gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
            o, f = e.lincomb(c.cid, k.reshape(-1), np.tile(gxy, n))
            w, wf = oracle_lib.batch_mul_base(c.cid, pyec.enc_scalar(c, scalars_to_int_sum(k.reshape(-1), L, c.n)))
            assert bytes(o) == bytes(w) and f == int(wf[0]), ("msm property", c.name, n, cb, dict(os.environ))
            stats["msm_property"] += 1
        else:
            P = pool(c)
            idx = np.array([rng.randrange(1 << 16) for _ in range(n)])
            pts = P[idx].copy()
            inf = np.zeros(n, np.uint8)
            for _ in range(rng.randrange(0, 6)):           # duplicates, cancelling pairs, identities, tiny / huge scalars
                i, j = rng.randrange(n), rng.randrange(n)
                what = rng.randrange(5)
                if what == 0: pts[j] = pts[i]; k[j] = k[i]
                elif what == 1:
                    pts[j] = pts[i]; k[j] = np.frombuffer(pyec.enc_scalar(c, (c.n - int.from_bytes(bytes(k[i]), "big")) % c.n), np.uint8)
                elif what == 2: inf[i] = 1; pts[i] = 0
                elif what == 3: k[i] = np.frombuffer(pyec.enc_scalar(c, rng.choice([0, 1, 2, c.n - 1, c.n - 2])), np.uint8)
                else: k[j] = k[i]
            o, f = e.lincomb(c.cid, k.reshape(-1), pts.reshape(-1), inf)
            w, wf = oracle_lib.msm(c.cid, k.reshape(-1), pts.reshape(-1), inf, vartime=True)
            assert bytes(o) == bytes(w) and f == wf, ("msm oracle", c.name, n, cb, dict(os.environ))
            stats["msm_oracle"] += 1
    elif kind == "fixed":
        n = rng.randrange(1, 3000)
        wdt = rng.choice([0, 0, rng.randrange(4, 17)])
        if wdt: e.set_base_window(c.cid, wdt)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        o, f = e.mul_by_generator(c.cid, k)
        w, wf = oracle_lib.batch_mul_base(c.cid, k)
        assert bytes(o) == bytes(w) and bytes(f) == bytes(wf), ("fixed", c.name, n, wdt)
        if wdt: e.set_base_window(c.cid, {"k256": 26, "p256": 24, "p384": 20, "sm2": 24, "p224": 24, "p192": 24, "p521": 20, "bp256": 24, "bp384": 20, "bp256t1": 24, "bp384t1": 20}[c.name])
        stats["fixed"] += 1
    else:
        n = rng.randrange(1, 1500)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        P = pool(c)
        pts = P[np.array([rng.randrange(1 << 16) for _ in range(n)])].copy().reshape(-1)
        o, f = e.mul(c.cid, k, pts)
        w, wf = oracle_lib.batch_mul(c.cid, k, pts)
        assert bytes(o) == bytes(w) and bytes(f) == bytes(wf), ("var", c.name, n)
        stats["var"] += 1
print("fuzz: all identical to the oracle / properties:", stats, "in %.0f s" % budget)
PY
