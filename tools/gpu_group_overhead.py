#!/usr/bin/env python3
"""What the single-process multi-GPU entry costs on top of the kernels it runs: ecgpu_group_msm_dev against
ecgpu_msm_dev on the same device and the same 2^21 k256 terms, wall clock per call (median and best of `reps`), for
  * a group of ONE member over RCCL (devices = [0]: local half, ncclAllGather in a one-rank communicator, combining half),
  * the same group after ecgpu_group_set_exchange(PEER) (the peer-copy leg),
  * a group of TWO members on one device (devices = [0, 0]: a worker thread, two local halves of 2^20 terms that share the GPU,
    two peer copies, one combining half) against two ecgpu_msm_dev calls of 2^20 terms back to back.
The group's result record is downloaded inside the call (65 bytes); ecgpu_msm_dev leaves its record on the device, so the single-
context side of the comparison includes a 65-byte ecgpu_copy_to_host as well.      python tools/gpu_group_overhead.py [log2 terms]"""
import importlib
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ec = importlib.import_module("elliptic-curves_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << lg
REPS = 40
e = ec.Engine(0)
g = torch.Generator(device="cuda")
g.manual_seed(21)
s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
s[:, 0] &= 0x7F
pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
k[:, 0] &= 0x7F
torch.cuda.synchronize()
e.mul_by_generator_dev(0, s, n, pts, None)
e.synchronize()
d_o = e.dev_alloc(64 + 64)


def single(m, parts):
    """`parts` back-to-back ecgpu_msm_dev calls of m terms each + the download of the last record"""
    ts = []
    out = None
    for _ in range(REPS):
        t0 = time.perf_counter()
        for j in range(parts):
            e.lincomb_dev(0, k[j * m:(j + 1) * m], pts[j * m:(j + 1) * m], None, m, d_o.at(0), d_o.at(64))
        out = bytes(e.to_host(d_o, 65))
        ts.append(time.perf_counter() - t0)
    return ts[3:], out


def group(devices, exchange=None):
    grp = ec.Group(devices, exchange=exchange)
    m = n // len(devices)
    ds = [k[j * m:(j + 1) * m] for j in range(len(devices))]
    dp = [pts[j * m:(j + 1) * m] for j in range(len(devices))]
    ts = []
    out = None
    try:
        for _ in range(REPS):
            t0 = time.perf_counter()
            out = grp.lincomb_dev(0, ds, dp, [m] * len(devices))
            ts.append(time.perf_counter() - t0)
        return ts[3:], out, grp.exchange
    finally:
        grp.close()


def show(label, ts):
    print("%-64s median %.3f ms  best %.3f ms" % (label, statistics.median(ts) * 1e3, min(ts) * 1e3), flush=True)
    return statistics.median(ts)


for round_ in range(2):          # twice: the second round is the one to read (clocks, allocations)
    t1, ref = single(n, 1)
    a = show("ecgpu_msm_dev, 2^%d terms (+ 65-byte download)" % lg, t1)
    t2, out, ex = group([0])
    b = show("ecgpu_group_msm_dev, devices [0], exchange %s" % ex, t2)
    assert bytes(out[0]) == ref[:64], "group result differs"
    print("    group - single: %+.1f us" % ((b - a) * 1e6))
    t3, out, ex = group([0], exchange="peer")
    c = show("ecgpu_group_msm_dev, devices [0], exchange %s" % ex, t3)
    assert bytes(out[0]) == ref[:64]
    print("    group - single: %+.1f us" % ((c - a) * 1e6))
    t4, _ = single(n // 2, 2)
    d = show("2 x ecgpu_msm_dev, 2^%d terms each, back to back" % (lg - 1), t4)
    t5, out, ex = group([0, 0])
    f = show("ecgpu_group_msm_dev, devices [0, 0] (2^%d terms each), %s" % (lg - 1, ex), t5)
    assert bytes(out[0]) == ref[:64]
    print("    group - 2 x single: %+.1f us   (one combining half instead of two, two local halves sharing the GPU)" % ((f - d) * 1e6))
e.close()
