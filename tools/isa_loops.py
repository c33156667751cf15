#!/usr/bin/env python3
"""The loops of a kernel on the gfx950 ISA: instructions per loop body and their histogram — the figures DESIGN.md quotes for the
serial chains ("193 instructions per doubling instead of 569" for k_msm_combine<K256Params>, "488 -> 353 instructions per batch" for
the inversion in k_normalize).  Compiles the kernel group's translation unit with -S (no GPU needed).

    python tools/isa_loops.py <group> <CurveParams> <mangled-name substring> [extra hipcc flags ...]
    python tools/isa_loops.py msm K256Params k_msm_combine
    python tools/isa_loops.py msm K256Params k_msm_combine -DECGPU_MSM_COMBINE_ROWS=0
    python tools/isa_loops.py base K256Params k_normalizeINS_10K256ParamsELi0E
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elliptic-curves_amd", "csrc")


def loops(group, curve, substr, flags=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DECGPU_CURVE=" + curve, "-S", "--offload-device-only",
                               "-o", out, os.path.join(CSRC, "ecgpu_inst_%s.hip" % group)] + list(flags), stderr=subprocess.DEVNULL)
        txt = open(out).read()
    res = []
    for m in re.finditer(r"^(_Z\w+):\s*;[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        if substr not in m.group(1):
            continue
        lines = [l.strip() for l in m.group(2).split("\n")]
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        body_all = [x for x in lines if x and not x.startswith((".", ";"))]
        found = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_cbranch\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
            if mm:
                t = mm.group(1) or mm.group(2)
                if t in labels and labels[t] < i:
                    body = [x for x in lines[labels[t]:i] if x and not x.startswith((".", ";"))]
                    found.append((t, len(body), collections.Counter(x.split()[0] for x in body)))
        res.append((m.group(1), len(body_all), found))
    return res


def main():
    if len(sys.argv) < 4:
        sys.exit(__doc__)
    group, curve, substr, flags = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    for name, total, found in loops(group, curve, substr, flags):
        print("%s  (%d instructions%s)" % (name, total, "; " + " ".join(flags) if flags else ""))
        for label, n, hist in found:
            print("  loop at %-12s %5d instructions  %s" % (label, n, ", ".join("%s %d" % kv for kv in hist.most_common(8))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
