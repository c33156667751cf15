#!/usr/bin/env python3
"""What a generator (comb) table of each width costs and buys on this GPU — the numbers behind the adaptive table policy
(include/ecgpu.h "the generator (comb) tables and their footprint", ecgpu_api.hip table_tier) and INTEGRATION.md's crossover
table.  Per curve and width: table bytes, build time, the first 1,024-scalar call on a cold width (build included), steady-state
milliseconds per 2^20-scalar batch (queued calls, drained inside the timed region).  Then the rent-or-buy thresholds those
numbers imply, and the default policy's first calls as a caller sees them.     python tools/gpu_table_tiers.py [curve ...]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ecgpu = importlib.import_module("elliptic-curves_amd")
dev = torch.device("cuda:0")
N = 1 << 20
K = 12


def scalars(n, L, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    s = torch.randint(0, 256, (n, L), dtype=torch.uint8, generator=g)
    s[:, 0] &= 0x7F
    return s.to(dev)


def steady(eng, cid, d_scal, out, inf):
    eng.set_async(True)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            eng.mul_by_generator_dev(cid, d_scal, N, out, inf)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        best = dt if best is None else min(best, dt)
    eng.set_async(False)
    return best


def main():
    curves = sys.argv[1:] or ["k256", "p256", "p384"]
    for name in curves:
        cid = ecgpu.CURVE_IDS[name]
        L = ecgpu.FIELD_BYTES[cid]
        wmax = {"k256": 26, "p384": 20, "p521": 20}.get(name, 24)
        d_small, d_big = scalars(1024, L, 1), scalars(N, L, 2)
        out, inf = torch.empty((N, 2 * L), dtype=torch.uint8, device=dev), torch.empty((N,), dtype=torch.uint8, device=dev)
        # the default (adaptive) policy as a caller meets it (first, while the device has seen nothing of this curve): a fresh
        # context, first calls of 1,024 and of 2^20 scalars, then the eager policy's first call
        eng = ecgpu.Engine(0)
        for n, d in ((1024, d_small), (N, d_big)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.mul_by_generator_dev(cid, d, n, out, inf)
            eng.synchronize()
            print("%s adaptive: call of %7d scalars %.2f ms, table now %s" % (name, n, (time.perf_counter() - t0) * 1e3, eng.base_table_info(cid)), flush=True)
        eng.set_table_policy(ecgpu.TABLE_EAGER)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.mul_by_generator_dev(cid, d_small, 1024, out, inf)
        eng.synchronize()
        print("%s eager:    call of %7d scalars %.2f ms, table now %s" % (name, 1024, (time.perf_counter() - t0) * 1e3, eng.base_table_info(cid)), flush=True)
        eng.close()
        rows = []
        for w in range(12, wmax + 1, 2):
            eng = ecgpu.Engine(0)
            eng.set_base_window(cid, w)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.mul_by_generator_dev(cid, d_small, 1024, out, inf)
            eng.synchronize()
            first_ms = (time.perf_counter() - t0) * 1e3
            info = eng.base_table_info(cid)
            eng.mul_by_generator_dev(cid, d_big, N, out, inf)
            eng.synchronize()
            ms = steady(eng, cid, d_big, out, inf)
            eng.close()
            rows.append((w, info["bytes"], info["build_ms"], first_ms, ms))
            print("%s w=%2d  table %9.1f MB  build %8.2f ms  first 1,024-scalar call %8.2f ms  steady %.4f ms per 2^20" % (
                name, w, info["bytes"] / 1e6, info["build_ms"], first_ms, ms), flush=True)
        # rent-or-buy: move from width a to width b once the time lost at a equals the build time of b
        by_w = {r[0]: r for r in rows}
        for a, b in ((16, 22), (22, wmax), (16, wmax)):
            if a in by_w and b in by_w and a < b and by_w[a][4] > by_w[b][4]:
                per_scalar_ns = (by_w[a][4] - by_w[b][4]) * 1e6 / N
                n_star = by_w[b][2] * 1e6 / per_scalar_ns
                print("%s  %d -> %d bits pays after %.3g scalars (2^%.1f): %.3f ns per scalar saved, %.1f ms to build" % (
                    name, a, b, n_star, __import__("math").log2(n_star), per_scalar_ns, by_w[b][2]), flush=True)


if __name__ == "__main__":
    main()
