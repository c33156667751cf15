#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against KNOWN bytes for the two access patterns the roofline `traffic` figures rest on:
  * random 64-byte gathers (the comb table of k_fixed_base, the point gathers of k_msm_accumulate): ecgpu_valu_probe(200),
    2^20 lanes x 16 entries = 1 GiB useful;
  * the per-lane tables of the variable-base kernels ([wave][entry][row][lane], 4 bytes per lane and load instruction):
    ecgpu_valu_probe(201) — every lane of a wave reads the same entry (contiguous 256-byte rows) — and (202) — a per-lane
    entry, as a digit-dependent ladder does; 524,288 lanes x 64 entry reads x 20 rows x 4 bytes = 2.5 GiB useful read,
    524,288 x 240 x 4 bytes = 480 MiB written.
Each probe runs in a process of its own under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, units of
the guide: FETCH_SIZE / WRITE_SIZE count kilobytes here, as tools/pmc_summary.py assumes for the bench passes).
    python tools/gpu_fetch_calibration.py            (parent)        --child N   (internal)"""
import csv
import glob
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
USEFUL = {200: ("k_gather_probe", (1 << 20) * 16 * 64, 0),
          201: ("k_tabrow_probe", 524288 * 64 * 20 * 4, 524288 * 240 * 4),
          202: ("k_tabrow_probe", 524288 * 64 * 20 * 4, 524288 * 240 * 4)}


def child(which):
    ec = importlib.import_module("elliptic-curves_amd")
    e = ec.Engine(0)
    rate = e.valu_probe(which)
    print("probe %d: %.1f GB/s of useful bytes" % (which, rate / 1e9))
    e.close()


def counter(which, name):
    out = "/tmp/fetch_cal_%d_%s" % (which, name)
    cmd = ["rocprofv3", "--pmc", name, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
           "--child", str(which)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    vals = []
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if USEFUL[which][0] in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    rate = [l for l in r.stdout.splitlines() if l.startswith("probe")]
    return (vals[-1] if vals else None), (rate[-1] if rate else r.stderr[-300:])


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    for which in (200, 201, 202):
        kern, rd, wr = USEFUL[which]
        f, rate = counter(which, "FETCH_SIZE")
        w, _ = counter(which, "WRITE_SIZE")
        print("%s" % rate)
        for label, val, useful in (("FETCH_SIZE", f, rd), ("WRITE_SIZE", w, wr)):
            if val is None:
                print("   %-10s no counter row" % label)
                continue
            as_kb = val * 1024
            print("   %-10s raw %.6g  -> as KiB: %.4g bytes = %.3f x the useful %s bytes (%.4g)" % (
                label, val, as_kb, as_kb / useful if useful else float("nan"), "read" if label[0] == "F" else "written", useful))


if __name__ == "__main__":
    main()
