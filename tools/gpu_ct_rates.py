#!/usr/bin/env python3
"""The uniform-schedule entry points on the GPU box: rates beside the variable-time kernels, and the dynamic twin of
tools/ct_isa_check.py — executed-instruction counters of the kernels for inputs that differ as much as inputs can.

    python tools/gpu_ct_rates.py               rates (2^20 device-resident elements, HIP-event kernel times), then the counters
    python tools/gpu_ct_rates.py --rates       the rates only
    python tools/gpu_ct_rates.py --child CLASS (internal) one launch set under rocprofv3 --pmc for input class CLASS

Counter check: each of {zero scalars, all-ones-pattern scalars (n - 1), random scalars} x {points = G, random points} is run
in a process of its own under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS`;
for k_var_base_ct / k_fixed_base_ct — and, for ecgpu_lincomb_ct, the tree of complete additions k_proj_sum_level over the
products (all its launches summed) — every counter must be IDENTICAL across the classes (no instruction is executed or
skipped depending on the data), while the variable-time kernels of the same inputs must differ (the check can see a
difference).  Sizes are fixed so that the launch geometry is the same."""
import csv
import glob
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib  # noqa: E402

CURVES = {"k256": 0, "p256": 1, "p384": 2}
LBYTES = {0: 32, 1: 32, 2: 48}
N_PMC = 1 << 14
CLASSES = ["zero_G", "nm1_G", "rand_G", "rand_rand", "zero_rand"]


def engine():
    ecgpu = importlib.import_module("elliptic-curves_amd")
    return ecgpu, ecgpu.Engine(0)


def rand_scalars(cid, n, seed):
    from gpu_common import rand_scalars as rs
    return rs(cid, n, seed)


def make_inputs(eng, ecgpu, cid, n, cls):
    L = LBYTES[cid]
    order = ecgpu.GROUP_ORDERS[cid]
    kind_k, kind_p = cls.split("_")
    if kind_k == "zero":
        ks = np.zeros(n * L, np.uint8)
    elif kind_k == "nm1":
        ks = np.frombuffer((order - 1).to_bytes(L, "big") * n, np.uint8).copy()
    else:
        ks = rand_scalars(cid, n, 0xC7A0 + cid)
    if kind_p == "G":
        one = np.frombuffer((1).to_bytes(L, "big") * n, np.uint8)
        pts, _ = eng.mul_by_generator(cid, one)
    else:
        pts, _ = eng.mul_by_generator(cid, rand_scalars(cid, n, 0xC7A1 + cid))
    return ks, pts


def child(cls):
    ecgpu, eng = engine()
    for name, cid in CURVES.items():
        L = LBYTES[cid]
        ks, pts = make_inputs(eng, ecgpu, cid, N_PMC, cls)
        d_k, d_p = eng.to_device(ks), eng.to_device(pts)
        d_o, d_f = eng.dev_alloc(N_PMC * 2 * L), eng.dev_alloc(N_PMC)
        for ct in (True, False):
            eng.mul_by_generator_dev(cid, d_k, N_PMC, d_o, d_f, constant_time=ct)
            eng.mul_dev(cid, d_k, d_p, None, N_PMC, d_o, d_f, constant_time=ct)
        eng.lincomb_ct_dev(cid, d_k, d_p, None, N_PMC, d_o, d_f)          # k_var_base_ct once more + the only k_proj_sum_level launches
    eng.close()


def counters(cls):
    out = "/tmp/ct_pmc_%s" % cls
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
           "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--child", cls]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    res = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ecgpu::", "")
            if not (k.startswith("k_var_base") or k.startswith("k_fixed_base") or k.startswith("k_proj_sum_level")):
                continue
            if k.startswith("k_proj_sum_level"):      # the levels of ecgpu_lincomb_ct's tree: all launches of the process, summed
                d = res.setdefault(k, {})
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                continue
            # the LAST launch of each kernel is the one on the class inputs (earlier ones build the point arrays)
            res.setdefault(k, {})[row["Counter_Name"]] = float(row["Counter_Value"])
    if not res:
        print("no counters for %s:\n%s" % (cls, r.stderr[-1500:]))
    return res


def rates():
    ecgpu, eng = engine()
    n = 1 << 20
    print("rates, 2^20 device-resident elements, kernel time from HIP events (main) and whole call (total):")
    for name, cid in CURVES.items():
        L = LBYTES[cid]
        ks = rand_scalars(cid, n, 0xC7B0 + cid)
        d_k = eng.to_device(ks)
        d_p, d_o, d_f = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n * 2 * L), eng.dev_alloc(n)
        eng.mul_by_generator_dev(cid, eng.to_device(rand_scalars(cid, n, 0xC7B1 + cid)), n, d_p, d_f)
        ref = {}
        for what in ("base", "var"):
            for ct in (False, True):
                for rep in range(3):
                    if what == "base":
                        eng.mul_by_generator_dev(cid, d_k, n, d_o, d_f, constant_time=ct)
                    else:
                        eng.mul_dev(cid, d_k, d_p, None, n, d_o, d_f, constant_time=ct)
                main, total = eng.last_timing("main"), eng.last_timing("total")
                got = bytes(eng.to_host(d_o, 4096 * 2 * L))
                if not ct:
                    ref[what] = (got, main)
                same = got == ref[what][0]
                print("  %-5s %-4s %-13s kernel %8.3f ms  call %8.3f ms  %10.4g /s  x%.2f of the variable-time kernel  results equal: %s" % (
                    name, what, "uniform" if ct else "variable-time", main, total, n / total * 1e3, main / ref[what][1], same), flush=True)
        # the constant-time `lincomb` beside the bucket method (variable-time names), whole calls
        for lg in (12, 16, 20):
            m = 1 << lg
            for rep in range(2):
                eng.lincomb_ct_dev(cid, d_k, d_p, None, m, d_o, d_f)
            t_ct = eng.last_timing("total")
            r_ct = bytes(eng.to_host(d_o, 2 * L))
            for rep in range(2):
                eng.lincomb_dev(cid, d_k, d_p, None, m, d_o, d_f)
            t_vt = eng.last_timing("total")
            print("  %-5s lincomb 2^%-2d terms: uniform schedule %8.3f ms (%.3g terms/s), bucket method %7.3f ms, results equal: %s" % (
                name, lg, t_ct, m / t_ct * 1e3, t_vt, r_ct == bytes(eng.to_host(d_o, 2 * L))), flush=True)
        for b in (d_k, d_p, d_o, d_f):
            b.free()
    eng.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    rates()
    if "--rates" in sys.argv:
        return 0
    print("\nexecuted-instruction counters per launch (%d elements), one process per input class:" % N_PMC)
    allc = {cls: counters(cls) for cls in CLASSES}
    kernels = sorted({k for c in allc.values() for k in c})
    bad = 0
    for k in kernels:
        rows = {cls: allc[cls].get(k, {}) for cls in CLASSES}
        names = sorted({c for r in rows.values() for c in r})
        uniform = all(len({rows[cls].get(c) for cls in CLASSES}) == 1 for c in names)
        is_ct = "_ct<" in k or k.startswith("k_proj_sum_level")
        verdict = "IDENTICAL" if uniform else "DIFFER"
        ok = uniform == is_ct
        bad += 0 if ok else 1
        print("  %-34s %s across %s  [%s]" % (k, verdict, "/".join(CLASSES), "as required" if ok else "UNEXPECTED"))
        for c in names:
            print("      %-18s %s" % (c, "  ".join("%.0f" % rows[cls].get(c, float("nan")) for cls in CLASSES)))
    print("counter check: %s" % ("PASS" if bad == 0 else "FAIL (%d kernels)" % bad))
    return bad


if __name__ == "__main__":
    sys.exit(main() or 0)
