#!/usr/bin/env python3
"""Build-time check of the wire-record codecs of the parameter sets whose records are not whole 16-byte vectors — p224
(28 / 56 bytes), p192 (24 / 48) and p521 (66 / 132) — on the gfx950 ISA.

Background (docs/DESIGN_long_form_r01-r05.md §4): p521's records were once decoded byte by byte; the compiler merged those loads into wide unaligned
ones and re-extracted the bytes with v_perm_b32 / SDWA sequences, and inside two large kernels the decoded operand had wrong
bits on gfx950.  load_wire / store_wire (csrc/ecgpu_kernels.h) now move whole 32-bit words (p521: one halfword + sixteen
words), which leaves nothing to extract.  This tool keeps it that way: it compiles the kernel groups to assembly (no GPU
needed) and asserts, per kernel,
  * byte loads (`global_load_ubyte` / `_sbyte`): at most 2 — the identity / parity / verdict flags, which ARE byte arrays —
    except in the message-hashing kernels, whose input is a byte string;
  * halfword loads: none for p224 / p192; for p521 at most one per wire record the kernel can read (<= 6);
  * no SDWA instruction with a byte selector on a loaded record (the extraction pattern) — allowed only in the hashing
    kernels and where a flag byte is widened (<= 2 per kernel);
  * record stores mirror the loads: byte stores <= 2 (flags), halfword stores <= 6 and only for p521.

    python tools/wire_codec_isa_check.py [--curve P224Params ...] [--groups base,var,msm,ct]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elliptic-curves_amd", "csrc")
HASHING = ("k_ecdsa_hash_msg", "k_sm2dsa_hash_msg", "k_schnorr_prepare_raw")


def kernels(asm):
    txt = open(asm).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        yield m.group(1), m.group(2)


def check(asm, curve):
    bad = 0
    for name, body in kernels(asm):
        c = collections.Counter()
        for line in body.splitlines():
            t = line.split(";")[0].strip()
            if not t:
                continue
            op = t.split()[0]
            if op.startswith(("global_load_ubyte", "global_load_sbyte", "flat_load_ubyte", "flat_load_sbyte")):
                c["byte loads"] += 1
            elif op.startswith(("global_load_ushort", "global_load_sshort", "global_load_short", "flat_load_ushort")):
                c["halfword loads"] += 1
            elif op.startswith("global_store_byte"):
                c["byte stores"] += 1
            elif op.startswith("global_store_short"):
                c["halfword stores"] += 1
            if "_sdwa" in op and "BYTE_" in t:
                c["sdwa byte selects"] += 1
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("void ecgpu::", "")
        hashing = any(h in name for h in HASHING)
        limits = {"byte loads": 2, "byte stores": 2, "sdwa byte selects": 2,
                  "halfword loads": 6 if curve == "P521Params" else 0, "halfword stores": 6 if curve == "P521Params" else 0}
        viol = [] if hashing else ["%s = %d (limit %d)" % (k, c[k], v) for k, v in limits.items() if c[k] > v]
        if viol:
            bad += 1
            print("  VIOLATION %-44s %s" % (short[:44], "; ".join(viol)))
        elif c:
            print("  ok        %-44s %s%s" % (short[:44], ", ".join("%s %d" % kv for kv in sorted(c.items())), "  (hashing kernel: byte input)" if hashing else ""))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", action="append")
    ap.add_argument("--groups", default="base,var,msm,ct")
    a = ap.parse_args()
    curves = a.curve or ["P224Params", "P192Params", "P521Params"]
    tmp = tempfile.mkdtemp(prefix="wire_isa_")
    jobs = []
    for c in curves:
        for g in a.groups.split(","):
            asm = os.path.join(tmp, "%s_%s.s" % (g, c))
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DECGPU_CURVE=" + c, "-S", "--offload-device-only",
                   os.path.join(CSRC, "ecgpu_inst_%s.hip" % g), "-o", asm]
            jobs.append((c, g, asm, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    bad = 0
    for c, g, asm, p in jobs:
        if p.wait() != 0:
            print("== %s, group %s: COMPILATION FAILED" % (c, g))
            bad += 1
            continue
        print("== %s, group %s" % (c, g))
        bad += check(asm, c)
        os.unlink(asm)
    os.rmdir(tmp)
    print("wire codec ISA check: %s" % ("PASS" if not bad else "%d kernels in violation" % bad))
    return bad


if __name__ == "__main__":
    sys.exit(min(main(), 255))
