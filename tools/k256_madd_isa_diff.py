#!/usr/bin/env python3
"""The mixed XYZZ addition of k_msm_accumulate<K256Params> on the gfx950 ISA: the compiler's rendering of the C++ field layer against the
build with the reduction as a hand-scheduled assembly block (-DECGPU_K256_ASM_REDUCE=1, csrc/ecgpu_k256_reduce_asm.h) — instruction
classes, issue slots, and the by-construction floor of tools/fixed_k256_floor.py beside them.  Cross-compiles (no GPU needed):
    python tools/k256_madd_isa_diff.py > profiles/r05/k256_madd_isa_diff.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "elliptic-curves_amd", "csrc", "ecgpu_inst_msm.hip")
HALF = {"v_and_b32", "v_and_b32_e32", "v_mov_b32", "v_mov_b32_e32", "v_add_u32", "v_add_u32_e32", "v_sub_u32_e32", "v_lshlrev_b32_e32",
        "v_lshrrev_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_cndmask_b32_e32"}


def madd_body(flags):
    """ISA lines of the xyzz_madd block of the accumulation loop: the largest basic-block run of the kernel (between the branch around
    the `fresh` path and its join)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-DECGPU_CURVE=K256Params",
                               "-S", "--cuda-device-only", SRC, "-o", out] + flags, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith("_ZN5ecgpu16k_msm_accumulate"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body, best, cur = lines[a:b], [], []
    for l in body:
        t = l.strip()
        if t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch"):
            if len(cur) > len(best):
                best = cur
            cur = []
        else:
            cur.append(t)
    return [l for l in best if l and not l.startswith(";")]


def hist(lines):
    h = collections.Counter(l.split()[0] for l in lines if l.split()[0].startswith(("v_", "s_")))
    valu = {k: v for k, v in h.items() if k.startswith("v_")}
    slots = sum(v * (0.5 if k in HALF else 1.0) for k, v in valu.items())
    return valu, slots


def main():
    cpp, asm = madd_body(["-DECGPU_K256_ASM_REDUCE=0"]), madd_body(["-DECGPU_K256_ASM_REDUCE=1"])
    hc, sc = hist(cpp)
    ha, sa = hist(asm)
    zero_ext = lambda ls: sum(1 for l in ls if re.match(r"v_mov_b32(_e32)? v\d+, (0|v\d+)$", l))
    print("k_msm_accumulate<K256Params>, the mixed XYZZ addition (8 products + 2 squares, 9 reductions, 3 folded differences) on gfx950")
    print("issue slots: VOP3 / 64-bit instruction = 1, 32-bit VOP1 / VOP2 = 0.5 (profiles/r01/isa_issue_rates.txt)\n")
    print("%-22s %12s %12s" % ("instruction", "C++ (compiler)", "asm reduce"))
    for k in sorted(set(hc) | set(ha), key=lambda k: -(hc.get(k, 0) + ha.get(k, 0))):
        print("%-22s %12d %12d" % (k, hc.get(k, 0), ha.get(k, 0)))
    print("%-22s %12d %12d" % ("VALU instructions", sum(hc.values()), sum(ha.values())))
    print("%-22s %12.1f %12.1f   (%.1f %% fewer)" % ("issue slots", sc, sa, 100 * (1 - sa / sc)))
    print("%-22s %12d %12d" % ("register moves", zero_ext(cpp), zero_ext(asm)))
    mads = 8 * 81 + 2 * 45 + 27
    print("\nby construction (tools/fixed_k256_floor.py): product columns + folded differences %d multiply-adds; a reduction 34 multiply-adds +" % mads)
    print("30 slots of carry pass and top fold (8 x (64-bit shift + 64-bit add + mask) + the last limb + three zero-extensions + the three")
    print("carries of the fold) = 64; linear steps of the formulas ~56: %d + 9 x 64 + 56 = %d slots." % (mads, mads + 9 * 64 + 56))
    print("\nwhere the compiler's %.0f extra slots per addition are (per reduction, x 9):" % (sc - sa))
    print("  * `col[j+1] += (u64)hi * G1` and `col[j+2] += (u64)hi * G2` (the upper half of a folded column) become 32-bit patches of the")
    print("    accumulators' UPPER words — v_mad_u32_u24 where the constant is visible, v_mov + v_mad_u64_u32 + v_mov where it is the opaque")
    print("    SGPR that keeps `x * 256` a multiply-add: 2 multiply-adds turn into ~4 slots;")
    print("  * the masked limbs r0..r2 are zero-extended through moves before the fold of the top column adds to them (v_mov_b32 vN, <zero>);")
    print("  * 12 64-bit adds and 12-13 masks per reduction against 10 + 12: the low halves of two columns are added before one shared")
    print("    multiply-add, which saves a multiply-add and costs an add + a move.")
    print("The assembly block keeps every column in its register pair from the last product multiply-add to the last carry, so none of")
    print("those moves exist; what it cannot do is overlap: the compiler interleaves the next product's multiply-adds with the serial carry")
    print("chain of a reduction, a block is scheduled as a unit (measured effect: profiles/r05/k256_asm_reduce_ab.txt).")


if __name__ == "__main__":
    main()
