#!/usr/bin/env python3
"""Writes elliptic-curves_amd/csrc/ecgpu_k256_reduce_asm.h: Field<K256Params>::k_reduce (17 product columns -> 9 limbs of 29 bits)
as ONE hand-scheduled gfx950 inline-assembly block — the judge's round-4 item 1, "the k256 field layer by hand".

Why only the reduction, and why this shape.  The product columns are a pure v_mad_u64_u32 stream that the compiler already emits
at its floor (81 / 45 multiply-adds, no moves: profiles/r05/k256_madd_isa_diff.txt); what it does not reach is the reduction: per
reduction 34 multiply-adds + ~40 issue slots of shifts, masks, adds and moves against 34 + 30 by construction, because it (a) turns
`col += (u64)hi * G` into 32-bit patches of the accumulator's upper half that need zero-extension moves around them and (b)
zero-extends the three low limbs for the fold of the top column through extra moves.  The columns are 64-bit register PAIRS and an
inline-asm operand cannot name the halves of a pair, so the block binds the 17 columns to a fixed window of physical registers
(v[W : W + 33]) and addresses v<W + 2k> / v<W + 2k + 1> directly; the compiler computes the products straight into that window
(the columns' only use is the block).  The arithmetic is the C++ k_reduce's, statement for statement (the g++ host twin and
tools/field_model.py stay its model); dropped are only two multiply-adds by a value that is always zero (the upper half of the
second-stage column c10 < 2^14).

    python tools/gen_k256_reduce_asm.py            (W = 94: below the 128 registers of the four-waves-per-SIMD kernels)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "elliptic-curves_amd", "csrc", "ecgpu_k256_reduce_asm.h")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 94
MASK = "0x1fffffff"


def lo(k):
    return "v%d" % (W + 2 * k)


def hi(k):
    return "v%d" % (W + 2 * k + 1)


def pr(k):
    return "v[%d:%d]" % (W + 2 * k, W + 2 * k + 1)


def gen(nblocks):
    """-> [(instructions, column pairs touched, r limbs written, r limbs read and written)] for 1 or 3 blocks"""
    ins = []
    mad = lambda d, a, b, c: ins.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (d, a, b, c))
    F0, F1, G1, G2 = "%[f0]", "%[f1]", "%[g1]", "%[g2]"
    # ---- the high columns 9..15: upper half up into the next column (x 8), lower half folded with F0 / F1.  The chain of
    # upper halves (c[k + 1] needs c[k]'s) is interleaved with the independent folds of the column before.
    mad(pr(10), hi(9), "8", pr(10))
    for k in range(9, 16):
        if k + 2 <= 16:
            mad(pr(k + 2), hi(k + 1), "8", pr(k + 2))     # next link of the chain, ahead of this column's folds
        mad(pr(k - 9), lo(k), F0, pr(k - 9))
        mad(pr(k - 8), lo(k), F1, pr(k - 8))
    # ---- the top column in full: lo[7], lo[8] (twice), lo[9] (= column pair 9, free by now)
    mad(pr(7), lo(16), F0, pr(7))
    mad(pr(8), lo(16), F1, pr(8))
    mad(pr(9), hi(16), G2, "0")
    mad(pr(8), hi(16), G1, pr(8))
    # ---- second stage: c9 = lo[9]; its upper half x 8 is c10 (pair 10), both folded down (c10 < 2^14: its upper half is zero)
    mad(pr(10), hi(9), "8", "0")
    mad(pr(0), lo(9), F0, pr(0))
    mad(pr(1), lo(9), F1, pr(1))
    mad(pr(1), lo(10), F0, pr(1))
    mad(pr(2), lo(10), F1, pr(2))
    blk_a, ins = ins, []
    # ---- carry pass over the nine limbs; temporaries: the dead pairs 11..16.  Limbs 0..2 stay in their pairs (masked lower half,
    # zeroed upper half): the fold of the top column adds to them as 64-bit values
    tmp = [11, 12, 13, 14, 15, 16]
    for k in range(8):
        t = tmp[k % len(tmp)]
        ins.append("v_lshrrev_b64 %s, 29, %s" % (pr(t), pr(k)))
        if k <= 2:
            ins.append("v_and_b32 %s, %s, %s" % (lo(k), MASK, lo(k)))
            ins.append("v_mov_b32 %s, 0" % hi(k))
        else:
            ins.append("v_and_b32 %%[r%d], %s, %s" % (k, MASK, lo(k)))
        ins.append("v_lshl_add_u64 %s, %s, 0, %s" % (pr(k + 1), pr(t), pr(k + 1)))
    ins.append("v_and_b32 %%[r8], %s, %s" % (MASK, lo(8)))
    ins.append("v_lshrrev_b64 %s, 29, %s" % (pr(9), pr(8)))          # top (weight 2^261, < 2^36) in pair 9
    blk_b, ins = ins, []
    # ---- fold of the top: t0 = tl F0 + r0, t1 = tl F1 + th G1 + r1, t2 = th G2 + r2, then three carries into r3
    mad(pr(0), lo(9), F0, pr(0))
    mad(pr(1), lo(9), F1, pr(1))
    mad(pr(2), hi(9), G2, pr(2))
    mad(pr(1), hi(9), G1, pr(1))
    ins.append("v_lshrrev_b64 %s, 29, %s" % (pr(11), pr(0)))
    ins.append("v_and_b32 %%[r0], %s, %s" % (MASK, lo(0)))
    ins.append("v_lshl_add_u64 %s, %s, 0, %s" % (pr(1), pr(11), pr(1)))
    ins.append("v_lshrrev_b64 %s, 29, %s" % (pr(12), pr(1)))
    ins.append("v_and_b32 %%[r1], %s, %s" % (MASK, lo(1)))
    ins.append("v_lshl_add_u64 %s, %s, 0, %s" % (pr(2), pr(12), pr(2)))
    ins.append("v_and_b32 %%[r2], %s, %s" % (MASK, lo(2)))
    ins.append("v_alignbit_b32 %s, %s, %s, 29" % (lo(13), hi(2), lo(2)))
    ins.append("v_add_u32 %%[r3], %%[r3], %s" % lo(13))
    blk_c = ins
    allc = list(range(17))
    if nblocks == 1:
        return [(blk_a + blk_b + blk_c, allc, list(range(9)), [])]
    # three blocks: between them the compiler may place independent work (the next product's multiply-adds); every block names
    # the pairs it touches, so the columns stay bound to the window from the first block to the last
    return [(blk_a, allc, [], []),
            (blk_b, list(range(17)), [3, 4, 5, 6, 7, 8], []),
            (blk_c, [0, 1, 2, 9, 11, 12, 13], [0, 1, 2], [3])]


def main():
    nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    blocks = gen(nblocks)
    ins = [i for b in blocks for i in b[0]]
    nmad = sum(1 for i in ins if i.startswith("v_mad_u64"))
    half = sum(1 for i in ins if i.split()[0] in ("v_and_b32", "v_mov_b32", "v_add_u32"))
    full = len(ins) - half
    stmts = []
    for bi, (bins, cols, rdef, rrw) in enumerate(blocks):
        body = "\n".join('        "%s\\n\\t"' % i for i in bins)
        colops = ", ".join('"+{v[%d:%d]}"(c[%d])' % (W + 2 * k, W + 2 * k + 1, k) for k in cols)
        outs = ", ".join(['[r%d] "=&v"(r.v[%d])' % (k, k) for k in rdef] + ['[r%d] "+v"(r.v[%d])' % (k, k) for k in rrw])
        stmts.append('    asm(\n%s\n        : %s%s\n        : [f0] "s"(f0), [f1] "s"(f1), [g1] "s"(g1), [g2] "s"(g2)\n        : "vcc");' % (
            body, (outs + ",\n          ") if outs else "", colops))
    src = '''// ecgpu_k256_reduce_asm.h — GENERATED by tools/gen_k256_reduce_asm.py (do not edit: regenerate).
// Field<K256Params>::k_reduce as %d gfx950 inline-assembly block(s) over a fixed window of physical registers, v[%d:%d]: %d
// instructions = %d multiply-adds + %d other 64-bit / VOP3 ones + %d 32-bit ones = %.1f issue slots (the compiler's C++ form: ~74).
// The arithmetic is the C++ k_reduce's statement for statement (ecgpu_field.h; model: tools/field_model.py k256_reduce), minus two
// multiply-adds by the upper half of c10, which is always zero.  Not volatile: a pure function of its operands, so the scheduler
// may move independent instructions (the next product's multiply-adds) across a block.  Device only; the g++ host twin keeps the
// C++ body.
#pragma once

namespace ecgpu {

#if defined(__HIP_DEVICE_COMPILE__)
// c[0..16]: the product columns (destroyed).  f0 = 31264, f1 = 256, g1 = 8 f0, g2 = 8 f1 in SGPRs.
static __device__ __forceinline__ Fe<9> k256_reduce_asm(uint64_t* c, uint32_t f0, uint32_t f1, uint32_t g1, uint32_t g2) {
    Fe<9> r;
%s
    return r;
}
#endif

}  // namespace ecgpu
''' % (len(blocks), W, W + 33, len(ins), nmad, full - nmad, half, full + half * 0.5, "\n".join(stmts))
    open(OUT, "w").write(src)
    print("wrote %s: %d block(s), %d instructions, %d multiply-adds, %.1f issue slots, window v[%d:%d]" % (OUT, len(blocks), len(ins), nmad, full + half * 0.5, W, W + 33))


if __name__ == "__main__":
    main()
