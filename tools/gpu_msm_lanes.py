#!/usr/bin/env python3
"""Do independent MSMs overlap usefully on one GPU?  One context runs K k256 MSMs back to back (synchronous calls); then two
contexts (two streams, two workspaces) run K each from two host threads; then ONE asynchronous context runs 2K with 1, 2, 3 and 4
MSM lanes (ecgpu_set_msm_lanes: the library's own form of the overlap).  Where the sort / tail phases of one MSM hide under the
accumulation of another, the time per MSM drops.    python tools/gpu_msm_lanes.py [log2 sizes ...]"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
from gpu_common import rand_scalars  # noqa: E402

cid = ec.CURVE_IDS["k256"]
L = 32
K = 12
for lg in [int(a) for a in sys.argv[1:]] or [21, 24]:
    n = 1 << lg
    engs = [ec.Engine(0), ec.Engine(0)]
    bufs = []
    for t, e in enumerate(engs):
        d_k = e.to_device(rand_scalars(cid, n, 0x2C0 + t))
        d_s = e.to_device(rand_scalars(cid, n, 0x2C8 + t))
        d_p, d_f = e.dev_alloc(n * 2 * L), e.dev_alloc(n)
        e.mul_by_generator_dev(cid, d_s, n, d_p, d_f)
        d_o, d_oi = e.dev_alloc(2 * L), e.dev_alloc(16)
        bufs.append((d_k, d_p, d_o, d_oi))

    def run(t, reps):
        e = engs[t]
        d_k, d_p, d_o, d_oi = bufs[t]
        for _ in range(reps):
            e.lincomb_dev(cid, d_k, d_p, None, n, d_o, d_oi)

    for t in (0, 1):
        run(t, 2)                                     # plans, workspaces, clocks
    t0 = time.perf_counter()
    run(0, K)
    one = (time.perf_counter() - t0) / K * 1e3
    ref = [bytes(engs[t].to_host(bufs[t][2], 2 * L)) for t in (0, 1)]
    th = [threading.Thread(target=run, args=(t, K)) for t in (0, 1)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    two = (time.perf_counter() - t0) / (2 * K) * 1e3
    same = all(bytes(engs[t].to_host(bufs[t][2], 2 * L)) == ref[t] for t in (0, 1))
    # one context, asynchronous, two MSM lanes (ecgpu_set_msm_lanes): the library's own form of the same overlap
    e = engs[0]
    d_k, d_p, d_o, d_oi = bufs[0]
    d_o2, d_oi2 = e.dev_alloc(2 * L), e.dev_alloc(16)
    outs = [(d_o, d_oi), (d_o2, d_oi2)] + [(e.dev_alloc(2 * L), e.dev_alloc(16)) for _ in range(2)]
    e.set_async(True)
    res = {}
    for nl in (1, 2, 3, 4):
        e.set_msm_lanes(nl)
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(2 * K):
                e.lincomb_dev(cid, d_k, d_p, None, n, *outs[i % 4])
            e.synchronize()
            res[nl] = (time.perf_counter() - t0) / (2 * K) * 1e3
        same = same and all(bytes(e.to_host(o[0], 2 * L)) == ref[0] for o in outs)
    e.set_msm_lanes(1)
    e.set_async(False)
    print("2^%d terms: one context, synchronous calls %.3f ms per MSM; two contexts side by side %.3f (x%.3f); one asynchronous context with "
          "1 / 2 / 3 / 4 MSM lanes %.3f / %.3f / %.3f / %.3f (x%.3f / %.3f / %.3f / %.3f); results unchanged: %s" % (
              lg, one, two, two / one, res[1], res[2], res[3], res[4], res[1] / one, res[2] / one, res[3] / one, res[4] / one, same), flush=True)
    for e in engs:
        e.close()
