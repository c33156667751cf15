#!/usr/bin/env python3
"""Model of the ROW-PARALLEL k256 field arithmetic of csrc/ecgpu_rows.h — one limb per lane, one field element per row of 16 lanes,
four independent products per wave — which k_msm_combine's Horner chain runs its doublings on (a serial chain of ~120 / ~240
complete doublings: one wave, so its time is its instruction count; a per-lane field multiplication is 145 instructions for the
lone wave, a row-parallel one ~75 with every cross-lane step a DPP row shift).

The model executes the kernel's statements on a row of 16 Python integers with the DPP semantics measured on gfx950
(tools/repro/lone_wave_latency.hip: `row_shr:n bound_ctrl:0` lane p <- lane p - n or 0, `row_shl:n bound_ctrl:0` lane p <- lane
p + n or 0, inside a row of 16), checks every intermediate against its register width (32 / 64 bits) and every result against
Python's integers mod p, on random and on adversarial operands (all limbs at the largest value the callers' magnitudes allow).

    python tools/rows_field_model.py            (no GPU)
"""
import random
import sys

P = 2 ** 256 - 2 ** 32 - 977
B = 29
M = (1 << B) - 1
F0, F1 = 31264, 256                    # 2^261 = F1 2^29 + F0 (mod p)
LB = 0x20100000                        # limb bound of a magnitude-1 element (ecgpu_field_consts.h K256U::LB)
Z1 = [0x3FFF820F, 0x3FFFFEF6] + [0x3FFFFFFE] * 6 + [0x20FFFFFE]     # 33 p, limbs in [LB, LB + 2^29)  (K256U::Z[1])
assert sum(z << (B * i) for i, z in enumerate(Z1)) == 33 * P
NPOS = 16


def chk(x, bits, what):
    assert 0 <= x < (1 << bits), "%s: %d bits needed, %d available" % (what, x.bit_length(), bits)
    return x


def shr(v, n):      # row_shr:n bound_ctrl:0
    return [v[p - n] if p - n >= 0 else 0 for p in range(NPOS)]


def shl(v, n):      # row_shl:n bound_ctrl:0
    return [v[p + n] if p + n < NPOS else 0 for p in range(NPOS)]


def value(v):
    return sum(x << (B * i) for i, x in enumerate(v))


def row(limbs):
    return list(limbs) + [0] * (NPOS - len(limbs))


# per-position constants (VGPRs in the kernel)
F0P = [F0 if p <= 9 else 0 for p in range(NPOS)]          # multiplier of the limb nine positions up
F1P = [F1 if 1 <= p <= 10 else 0 for p in range(NPOS)]    # multiplier of the limb eight positions up (position 0: limb 8 is not high)
LOW9 = [1 if p <= 8 else 0 for p in range(NPOS)]


def mul_rows(a, b, stats=None):
    """a, b: rows with limbs at positions 0..8 (lazy: limb magnitudes ma, mb with ma mb <= 7), zeros above -> the product, limbs < LB"""
    assert all(x == 0 for x in a[9:]) and all(x == 0 for x in b[9:])
    # --- 16 product columns: c_p = sum_i a_i b_(p - i); a_i arrives through the LDS crossbar (a broadcast inside the row), b shifted
    # by i positions is one DPP move
    c = [0] * NPOS
    for i in range(9):
        bs = shr(b, i) if i else b
        for p in range(NPOS):
            c[p] = chk(c[p] + a[i] * bs[p], 64, "column")
    top = chk(a[8] * b[8], 64, "top column")                 # column 16: every lane of the row computes it
    # --- stage 1: every column in three pieces of 29 / 29 / 6 bits, each added where it weighs 2^(29 p)
    l = [x & M for x in c]
    m = [(x >> B) & M for x in c]
    h = [x >> (2 * B) for x in c]
    c1 = [chk(l[p] + shr(m, 1)[p] + shr(h, 2)[p], 32, "stage 1") for p in range(NPOS)]      # positions 0..15
    # positions 16, 17, 18 live in a second register at positions 0, 1, 2
    tl, tm, th = top & M, (top >> B) & M, top >> (2 * B)
    t = [0] * NPOS
    m15, h14 = shl(m, 15), shl(h, 14)                          # lane 0 <- m_15 | lane 0 <- h_14, lane 1 <- h_15
    for p in range(NPOS):
        t[p] = chk(m15[p] + h14[p] + (tl if p == 0 else tm if p == 1 else th if p == 2 else 0), 32, "stage 1 top")
    # --- stage 2: the limbs at positions 9..18 folded down: position j takes F0 * limb (j + 9) + F1 * limb (j + 8)
    ha = [shl(c1, 9)[p] + shr(t, 7)[p] for p in range(NPOS)]      # limb p + 9: lanes 0..6 from c1, lanes 7..9 from t
    hb = [shl(c1, 8)[p] + shr(t, 8)[p] for p in range(NPOS)]      # limb p + 8: lanes 0..7 from c1 (lane 0: limb 8, multiplier 0), 8..10 from t
    r = [chk(c1[p] * LOW9[p] + F0P[p] * ha[p] + F1P[p] * hb[p], 64, "stage 2") for p in range(NPOS)]
    if stats is not None:
        stats["stage2"] = max(stats.get("stage2", 0), max(r))
    # --- stage 3: carry pass (two pieces: the values are < 2^46)
    l = [x & M for x in r]
    m = [chk(x >> B, 32, "stage 3 carry") for x in r]
    r1 = [chk(l[p] + shr(m, 1)[p], 32, "stage 3") for p in range(NPOS)]         # positions 0..11
    # --- stage 4: positions 9, 10 (and 11: zero) folded down once more
    ga, gb = shl(r1, 9), shl(r1, 8)
    r2 = [chk(r1[p] * LOW9[p] + F0P[p] * ga[p] + F1P[p] * gb[p], 64, "stage 4") for p in range(NPOS)]
    assert all(x == 0 for x in r2[9:]), r2
    # --- stage 5: carries of positions 0, 1, 2 only (the others are below 2^29 + 2^17 already and stay as they are)
    l = [(x & M) if p <= 2 else x for p, x in enumerate(r2)]
    m = [(x >> B) if p <= 2 else 0 for p, x in enumerate(r2)]
    out = [chk(l[p] + shr(m, 1)[p], 32, "stage 5") for p in range(NPOS)]
    assert all(x == 0 for x in out[9:])
    assert all(x < LB for x in out[:9]), [hex(x) for x in out]
    if stats is not None:
        stats["out"] = max(stats.get("out", 0), max(out))
    return out


def norm64_rows(v):
    """v: 64-bit per-position values at positions 0..8 (a small linear combination of magnitude-1 elements, < 2^38 per limb)
    -> limbs < 2 LB (magnitude 2)"""
    assert all(x == 0 for x in v[9:])
    l = [x & M for x in v]
    m = [chk(x >> B, 32, "norm carry") for x in v]
    v1 = [chk(l[p] + shr(m, 1)[p], 32, "norm") for p in range(NPOS)]            # positions 0..9
    ga, gb = shl(v1, 9), shl(v1, 8)
    out = [chk(v1[p] * LOW9[p] + F0P[p] * ga[p] + F1P[p] * gb[p], 32, "norm fold") for p in range(NPOS)]
    assert all(x == 0 for x in out[9:])
    assert all(x < 2 * LB for x in out[:9]), [hex(x) for x in out]
    return out


B3 = 21                                   # 3 b, b = 7


def dbl_rows(X, Y, Z):
    """the complete doubling (Renes-Costello-Batina 2016 algorithm 9, a = 0) in the kernel's order: level 1 = four products on the
    four rows, the small multiples and their normalisation, level 2 = four products.  X, Y, Z: rows (magnitudes <= 2, 2, 1)."""
    add = lambda a, b: [chk(x + y, 32, "add") for x, y in zip(a, b)]
    A, Bq = [Y, Y, Z, X], [Y, Z, Z, Y]
    Pr = [mul_rows(A[r], Bq[r]) for r in range(4)]             # Y^2 | Y Z | Z^2 | X Y
    T0, T2 = Pr[0], Pr[2]
    big = row(Z1)
    nT2 = [chk(big[p] - T2[p], 32, "bias") for p in range(NPOS)]                # 33 p - Z^2, limb-wise non-negative
    a0, a2, a2n, aP = [0, 0, 1, 1], [B3, 0, 0, 0], [0, 0, 3 * B3, 3 * B3], [0, 1, 0, 0]
    b0, b2, bP = [8, 8, 1, 0], [0, 0, B3, 0], [0, 0, 0, 2]
    A2 = [norm64_rows([chk(a0[r] * T0[p] + a2[r] * T2[p] + a2n[r] * nT2[p] + aP[r] * Pr[r][p], 64, "lin A") for p in range(NPOS)]) for r in range(4)]
    B2 = [norm64_rows([chk(b0[r] * T0[p] + b2[r] * T2[p] + bP[r] * Pr[r][p], 64, "lin B") for p in range(NPOS)]) for r in range(4)]
    Q = [mul_rows(A2[r], B2[r]) for r in range(4)]
    return Q[3], add(Q[0], Q[2]), Q[1]


def ref_dbl(x, y, z):
    t0 = y * y % P
    z3 = 8 * t0 % P
    t1 = y * z % P
    t2 = B3 * z * z % P
    x3 = t2 * z3 % P
    y3 = (t0 + t2) % P
    z3 = t1 * z3 % P
    t0 = (t0 - 3 * t2) % P
    y3 = (x3 + t0 * y3) % P
    x3 = 2 * t0 * (x * y % P) % P
    return x3, y3, z3


def main():
    rnd = random.Random(5)
    stats = {}
    n = 0
    for mag_a, mag_b in ((1, 1), (2, 2), (2, 1), (1, 7), (7, 1), (2, 3), (3, 2)):
        for trial in range(400):
            if trial == 0:
                a = row([mag_a * LB - 1] * 9)
                b = row([mag_b * LB - 1] * 9)
            elif trial == 1:
                a, b = row([0] * 9), row([mag_b * LB - 1] * 9)
            else:
                a = row([rnd.randrange(mag_a * LB) if rnd.random() < 0.8 else mag_a * LB - 1 for _ in range(9)])
                b = row([rnd.randrange(mag_b * LB) if rnd.random() < 0.8 else mag_b * LB - 1 for _ in range(9)])
            r = mul_rows(a, b, stats)
            assert value(r) % P == value(a) * value(b) % P
            n += 1
    print("mul_rows: %d products equal to the integers mod p; limb magnitudes up to 7 x 1; largest stage-2 value 2^%.1f, largest output limb 2^29 + 2^%.1f (LB = 2^29 + 2^20)" % (
        n, __import__("math").log2(stats["stage2"]), __import__("math").log2(stats["out"] - (1 << 29))))
    n = 0
    for trial in range(300):
        lim = [2 * LB, 2 * LB, LB]
        if trial == 0:
            X, Y, Zr = (row([l - 1] * 9) for l in lim)
        else:
            X, Y, Zr = (row([rnd.randrange(l) if rnd.random() < 0.8 else l - 1 for _ in range(9)]) for l in lim)
        for step in range(4):                                   # a chain: the outputs are the next inputs
            x3, y3, z3 = dbl_rows(X, Y, Zr)
            assert (value(x3) % P, value(y3) % P, value(z3) % P) == ref_dbl(value(X) % P, value(Y) % P, value(Zr) % P)
            X, Y, Zr = x3, y3, z3
            n += 1
    print("dbl_rows: %d doublings (chains of 4) equal to the complete doubling formulas mod p; every intermediate inside its register" % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
