#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (tools/gpu_run.sh recipes `prof:` and `pmc:`) into small text / JSON summaries.

    pmc_summary.py stats <dir>          per-kernel calls / average ms from *kernel_stats.csv
    pmc_summary.py pmc <outdir> <tag>   per-kernel counter averages over every <outdir>/pmc_<tag>_*/ pass, plus
                                        <outdir>/pmc_<tag>.json = {kernel: {counter: average per launch}}
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    # (k_var_base<C, false> / <C, true>: the plain ladder and the one that adds its product to a G — one name for both, as bench.py
    # and roofline_consts.py key them; a workload's profile holds the variant that workload runs)
    return name.split("(")[0].replace("void ", "").replace("ecgpu::", "").replace(", false>", ">").replace(", true>", ">")


def stats(d):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ecgpu" in r["Name"]:
                print("%-58s calls=%-4s avg_ms=%.4f total_ms=%.3f" % (short(r["Name"])[:58], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                                       float(r["TotalDurationNs"]) / 1e6))


def pmc(outdir, tag):
    acc = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(outdir, "pmc_%s_*" % tag))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if "probe" in k or "ecgpu" not in r["Kernel_Name"]:
                    continue
                acc[(k, r["Counter_Name"])].append((int(r.get("Grid_Size") or 0), float(r["Counter_Value"])))
    summary = collections.defaultdict(dict)
    for (k, c), rows in sorted(acc.items()):
        # a kernel also runs at other sizes in a process (table construction, the one-point normalisation of the bench's
        # check): the timed launches are the ones of the most frequent grid size (ties: the largest) — average those
        sizes = collections.Counter(g for g, _ in rows)
        grid = max(sizes, key=lambda g: (sizes[g], g))
        v = [x for g, x in rows if g == grid]
        summary[k][c] = sum(v) / len(v)
        print("%-58s %-22s n=%-3d grid=%-9d avg=%.6g" % (k[:58], c, len(v), grid, summary[k][c]))
    with open(os.path.join(outdir, "pmc_%s.json" % tag), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
