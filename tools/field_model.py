#!/usr/bin/env python3
"""Exact integer models of the unsaturated-limb field multiplications used on the GPU, with every
64-bit accumulator checked for overflow.  Used (a) to validate the bound analysis in DESIGN.md §3 with
adversarial (maximal-limb) inputs, and (b) by tools/gen_field_consts.py to derive constants.

k256:  9 limbs x 29 bits, plain residues, value < M * 2^261, fold with 2^261 = 2^37 + 31264 (mod p)
p256: 10 limbs x 28 bits, Montgomery form R = 2^280, p' = 1 (p = -1 mod 2^28), value < M * 2p
"""
import random

U64 = (1 << 64) - 1


def chk64(x):
    assert 0 <= x <= U64, "64-bit accumulator overflow: %x" % x
    return x


def chk32(x):
    assert 0 <= x < (1 << 32), "32-bit limb overflow: %x" % x
    return x


def to_limbs(v, nl, b):
    return [(v >> (b * i)) & ((1 << b) - 1) for i in range(nl - 1)] + [v >> (b * (nl - 1))]


def from_limbs(l, b):
    return sum(x << (b * i) for i, x in enumerate(l))


# ------------------------------------------------------------------------------------------------
# k256, 9 x 29
# ------------------------------------------------------------------------------------------------
K_P = 2 ** 256 - 2 ** 32 - 977
K_NL, K_B = 9, 29
K_MASK = (1 << K_B) - 1
K_F0, K_F1 = 31264, 256            # 2^261 = 2^5 * (2^32 + 977) = 256 * 2^29 + 31264
K_G1, K_G2 = 8 * 31264, 8 * 256    # 2^(261+32) = 2^3 * 2^29 * 2^261: hi halves of 64-bit columns
K_LB = (1 << 29) + (1 << 20)       # limb bound of a magnitude-1 element (small slack over 2^29)


def k256_columns(a, b):
    c = [0] * 17
    for i in range(9):
        for j in range(9):
            c[i + j] = chk64(c[i + j] + chk32(a[i]) * chk32(b[j]))
    return c


def k256_reduce(c):
    """17 64-bit columns -> 9 limbs, magnitude 1 (limbs < K_LB, value congruent mod p)."""
    lo = c[:9] + [0, 0]
    c = list(c)

    def fold_full(j, col):                     # both 32-bit halves of a column of weight 2^(261 + 29 j): four multiply-adds
        cl, ch = col & 0xFFFFFFFF, col >> 32
        lo[j] = chk64(lo[j] + cl * K_F0)
        lo[j + 1] = chk64(lo[j + 1] + cl * K_F1 + ch * K_G1)
        lo[j + 2] = chk64(lo[j + 2] + ch * K_G2)

    for k in range(9, 16):                     # columns 9..15: the upper half moves into the next column (2^32 = 8 * 2^29),
        cl, ch = c[k] & 0xFFFFFFFF, c[k] >> 32  # the lower half folds with two multiply-adds
        c[k + 1] = chk64(c[k + 1] + ch * 8)
        j = k - 9
        lo[j] = chk64(lo[j] + cl * K_F0)
        lo[j + 1] = chk64(lo[j + 1] + cl * K_F1)
    fold_full(7, c[16])
    # columns 9 and 10 (weights 2^261, 2^290) are folded once more, the same way
    c9, c10 = lo[9], lo[10]
    cl, ch = c9 & 0xFFFFFFFF, c9 >> 32
    c10 = chk64(c10 + ch * 8)
    lo[0] = chk64(lo[0] + cl * K_F0)
    lo[1] = chk64(lo[1] + cl * K_F1)
    lo[9] = lo[10] = 0                         # (fold_full below may touch index 3 at most)
    fold_full(1, c10)
    lo = lo[:9]
    # sequential carry propagation 0 -> 8
    r = [0] * 9
    carry = 0
    for k in range(9):
        v = chk64(lo[k] + carry)
        if k < 8:
            r[k] = v & K_MASK
            carry = v >> K_B
        else:
            r[8] = v & K_MASK
            top = v >> K_B                      # weight 2^261, < 2^36
    # fold the top once more (halves), then a short carry 0 -> 3
    tl, th = top & 0xFFFFFFFF, top >> 32
    t0 = chk64(r[0] + tl * K_F0)
    t1 = chk64(r[1] + tl * K_F1 + th * K_G1)
    t2 = chk64(r[2] + th * K_G2)
    r[0] = t0 & K_MASK
    t1 = chk64(t1 + (t0 >> K_B))
    r[1] = t1 & K_MASK
    t2 = chk64(t2 + (t1 >> K_B))
    r[2] = t2 & K_MASK
    r[3] = chk32(r[3] + (t2 >> K_B))           # no further propagation: slack absorbed by K_LB
    assert all(x < K_LB for x in r), [hex(x) for x in r]
    return r


def k256_mul(a, b):
    return k256_reduce(k256_columns(a, b))


def k256_zconst(m):
    """Z[m] of the generated constants (sub_constant below: what tools/gen_field_consts.py writes into K256U::Z)."""
    return sub_constant(K_P, 9, 29, m, K_LB, K_LB)[0]


def k256_mul_sub(a, b, c, m):
    """a * b - c with one reduction (Field::mul_sub): (Z[m] - c) enters the low columns; c has magnitude m."""
    col = k256_columns(a, b)
    z = k256_zconst(m)
    for k in range(9):
        d = chk32(z[k] - c[k])
        assert z[k] >= c[k]
        col[k] = chk64(col[k] + d)
    return k256_reduce(col)


def k256_to_words_m1(a):
    """Field::k_to_words<NORMED = true>: the canonical value of a magnitude-1 element (limbs < K_LB) WITHOUT the two carry
    passes a lazy element needs first — two folds of everything at or above 2^256, one carry pass, one conditional subtraction."""
    r = list(a)
    assert all(x < K_LB for x in r)
    for _ in range(2):
        carry = 0
        for k in range(8):
            v = chk32(r[k] + carry)
            r[k] = v & K_MASK
            carry = v >> K_B
        v8 = chk32(r[8] + carry)
        e = v8 >> 24
        r[8] = v8 & 0xFFFFFF
        r[0] = chk32(r[0] + e * 977)
        r[1] = chk32(r[1] + e * 8)
    carry = 0
    for k in range(8):
        v = r[k] + carry
        r[k] = v & K_MASK
        carry = v >> K_B
    r[8] += carry
    s = list(r)
    s[0] += 977
    s[1] += 8
    carry = 0
    for k in range(8):
        v = s[k] + carry
        s[k] = v & K_MASK
        carry = v >> K_B
    s[8] += carry
    ge = (s[8] >> 24) != 0
    s[8] &= 0xFFFFFF
    out = s if ge else r
    assert all(x <= K_MASK for x in out[:8]) and out[8] < (1 << 24)
    return from_limbs(out, K_B)


def k256_norm(a):
    """carry-propagate a lazy element (limbs < 2^32) and fold the top: magnitude 1."""
    r = [0] * 9
    carry = 0
    for k in range(9):
        v = chk32(a[k]) + carry
        r[k] = v & K_MASK
        carry = v >> K_B
    # carry has weight 2^261, < 2^4
    r[0] = chk32(r[0] + carry * K_F0)
    r[1] = chk32(r[1] + carry * K_F1)
    assert all(x < K_LB for x in r)
    return r


# ------------------------------------------------------------------------------------------------
# p256, 10 x 28, Montgomery R = 2^280
# ------------------------------------------------------------------------------------------------
P_P = 2 ** 256 - 2 ** 224 + 2 ** 192 + 2 ** 96 - 1
P_NL, P_B = 10, 28
P_MASK = (1 << P_B) - 1
P_R = 1 << (P_NL * P_B)
P_LIMBS = to_limbs(P_P, P_NL, P_B)
assert P_LIMBS[0] == P_MASK                      # p = -1 mod 2^28  =>  p' = 1
P_LB = (1 << 28) + (1 << 19)


def p256_reduce_rows(c):
    """The 10 Montgomery rows of p256 in the sparse form the kernels use:
    u p = -u + u 2^96 + u 2^192 + u 2^224 (2^32 - 1), i.e. in 28-bit columns relative to row i: the -u clears the low
    28 bits of c[i] (leaving the carry c[i] >> 28 for column i+1), + u 2^12 into column i+3, + u 2^24 into column
    i+6, + u (2^32 - 1) into column i+8.  3 multiply-adds per row instead of 6 for the limb form of p; the last one
    adds up to 2^60 to a column, which is why the product limit of p256 is 23 and not 24."""
    for i in range(10):
        u = c[i] & P_MASK
        c[i + 1] = chk64(c[i + 1] + (c[i] >> P_B))
        c[i + 3] = chk64(c[i + 3] + u * (1 << 12))
        c[i + 6] = chk64(c[i + 6] + u * (1 << 24))
        c[i + 8] = chk64(c[i + 8] + u * 0xFFFFFFFF)


def p256_mont_mul(a, b, c_extra=None, hi_add=None):
    c = [0] * 21
    for i in range(10):
        for j in range(10):
            c[i + j] = chk64(c[i + j] + chk32(a[i]) * chk32(b[j]))
    if c_extra is not None:                      # mul2: a second product shares the column accumulators
        x, y = c_extra
        for i in range(10):
            for j in range(10):
                c[i + j] = chk64(c[i + j] + chk32(x[i]) * chk32(y[j]))
    if hi_add is not None:                       # mul_sub / sqr_sub: (multiple of p) - c enters the result columns
        for k in range(10):
            c[10 + k] = chk64(c[10 + k] + chk32(hi_add[k]))
    p256_reduce_rows(c)
    r = [0] * 10
    carry = 0
    for k in range(10):
        v = chk64(c[10 + k] + carry)
        if k < 9:
            r[k] = v & P_MASK
            carry = v >> P_B
        else:
            r[9] = chk32(v)
    assert all(x < (1 << 28) for x in r[:9]) and (hi_add is not None or r[9] < 32), [hex(x) for x in r]
    return r


def p256_norm(a):
    r = [0] * 10
    carry = 0
    for k in range(10):
        v = chk32(a[k]) + carry
        if k < 9:
            r[k] = v & P_MASK
            carry = v >> P_B
        else:
            r[9] = chk32(v)
    return r


# ------------------------------------------------------------------------------------------------
# p384, 15 x 27, Montgomery R = 2^405 (same algorithm as p256, generic in NL / B)
# ------------------------------------------------------------------------------------------------
Q_P = 2 ** 384 - 2 ** 128 - 2 ** 96 + 2 ** 32 - 1
Q_NL, Q_B = 15, 27
Q_MASK = (1 << Q_B) - 1
Q_R = 1 << (Q_NL * Q_B)
Q_LIMBS = to_limbs(Q_P, Q_NL, Q_B)
assert Q_LIMBS[0] == Q_MASK                      # p = -1 mod 2^27  =>  p' = 1
Q_LB = (1 << 27) + (1 << 18)
Q_TOP1 = 128                                     # top limb (bits 378..) of a value < 2p


def umont_mul(a, b, nl, bits, plimbs, hi_add=None):
    """generic unsaturated Montgomery multiplication (p = -1 mod 2^bits), every accumulator checked"""
    mask = (1 << bits) - 1
    c = [0] * (2 * nl + 1)
    for i in range(nl):
        for j in range(nl):
            c[i + j] = chk64(c[i + j] + chk32(a[i]) * chk32(b[j]))
    if hi_add is not None:
        for k in range(nl):
            c[nl + k] = chk64(c[nl + k] + chk32(hi_add[k]))
    for i in range(nl):
        u = c[i] & mask
        c[i + 1] = chk64(c[i + 1] + (c[i] >> bits) + u * (plimbs[1] + 1))
        for j in range(2, nl):
            if plimbs[j]:
                c[i + j] = chk64(c[i + j] + u * plimbs[j])
    r = [0] * nl
    carry = 0
    for k in range(nl):
        v = chk64(c[nl + k] + carry)
        if k < nl - 1:
            r[k] = v & mask
            carry = v >> bits
        else:
            r[k] = chk32(v)
    return r


P224_P = 2 ** 224 - 2 ** 96 + 1                  # p224/src/arithmetic/field.rs:54-61


def umont_mul_general(a, b, nl, bits, plimbs, c_extra=None, hi_add=None):
    """unsaturated Montgomery multiplication for any odd p: u = c_i * (-p^-1) mod 2^bits, c += u * p, every accumulator
    checked.  For p224, p = 1 mod 2^28, so -p^-1 = -1 and u = -c_i mod 2^28."""
    mask = (1 << bits) - 1
    pinv = (-pow(from_limbs(plimbs, bits), -1, 1 << bits)) % (1 << bits)
    c = [0] * (2 * nl + 1)
    for i in range(nl):
        for j in range(nl):
            c[i + j] = chk64(c[i + j] + chk32(a[i]) * chk32(b[j]))
    if c_extra is not None:
        for i in range(nl):
            for j in range(nl):
                c[i + j] = chk64(c[i + j] + chk32(c_extra[0][i]) * chk32(c_extra[1][j]))
    if hi_add is not None:
        for k in range(nl):
            c[nl + k] = chk64(c[nl + k] + chk32(hi_add[k]))
    for i in range(nl):
        u = (c[i] * pinv) & mask
        for j in range(nl):
            if plimbs[j]:
                c[i + j] = chk64(c[i + j] + u * plimbs[j])
        assert c[i] & mask == 0
        c[i + 1] = chk64(c[i + 1] + (c[i] >> bits))
    r = [0] * nl
    carry = 0
    for k in range(nl):
        v = chk64(c[nl + k] + carry)
        if k < nl - 1:
            r[k] = v & mask
            carry = v >> bits
        else:
            r[k] = chk32(v)
    return r


# ------------------------------------------------------------------------------------------------
# subtraction constants: a multiple of p whose limbs all dominate a magnitude-M element
# ------------------------------------------------------------------------------------------------
def sub_constant(p, nl, b, m, lb, top_bound):
    """limbs l_j in [m*lb, m*lb + 2^b) for j < nl-1, top limb >= m*top_bound, sum = 0 mod p."""
    base = sum((m * lb) << (b * j) for j in range(nl - 1)) + ((m * top_bound) << (b * (nl - 1)))
    k = -(-base // p)
    rest = k * p - base
    d = to_limbs(rest, nl, b)
    limbs = [d[j] + m * lb for j in range(nl - 1)] + [d[nl - 1] + m * top_bound]
    assert from_limbs(limbs, b) == k * p and all(x < (1 << 32) for x in limbs)
    return limbs, k


def chk_s64(x):
    assert -(1 << 63) <= x < (1 << 63), "signed 64-bit accumulator overflow: %x" % x
    return x


def p384_mont_mul(a, b, c_extra=None, hi_add=None):
    """p384 with SIGNED column accumulators and the Montgomery rows in sparse form:
    u p = -u + u 2^32 - u 2^96 - u 2^128 + u 2^384; in 27-bit columns relative to row i: the -u clears the low 27
    bits of c[i] (carry = arithmetic c[i] >> 27 into column i+1), + u 2^5 into column i+1, - u 2^15 into column i+3,
    - u 2^20 into column i+4, + u 2^6 into column i+14.  4 multiply-adds per row instead of 13 for the limb form of
    p; the sign bit costs one bit of headroom, which is why the product limit of p384 is 30 and not 60."""
    c = [0] * 31
    for x, y in ((a, b),) + ((c_extra,) if c_extra is not None else ()):
        for i in range(15):
            for j in range(15):
                c[i + j] = chk_s64(c[i + j] + chk32(x[i]) * chk32(y[j]))
    if hi_add is not None:
        for k in range(15):
            c[15 + k] = chk_s64(c[15 + k] + chk32(hi_add[k]))
    for i in range(15):
        u = c[i] & Q_MASK
        c[i + 1] = chk_s64(c[i + 1] + (c[i] >> Q_B) + u * (1 << 5))
        c[i + 3] = chk_s64(c[i + 3] - u * (1 << 15))
        c[i + 4] = chk_s64(c[i + 4] - u * (1 << 20))
        c[i + 14] = chk_s64(c[i + 14] + u * (1 << 6))
    r = [0] * 15
    v = c[15]
    for k in range(14):
        r[k] = v & Q_MASK
        v = chk_s64(c[16 + k] + (v >> Q_B))
    r[14] = chk32(v)
    return r


# ------------------------------------------------------------------------------------------------
# sparse Montgomery rows on UNSIGNED columns with a bias (sm2, p224, p192)
# ------------------------------------------------------------------------------------------------
# u p is written as a few power-of-two terms; the negative ones would take an unsigned column below zero, so every column that
# receives one starts from a bias instead of zero, and the biases together are a multiple k p of the modulus (plus its
# canonical rest in the low limbs): the value being reduced is a b + k p, the result is unchanged mod p, and k p / R is
# negligible against p.  SPARSE[name] = (p, limbs, bits, p0 is one?, [(column offset, coefficient)], bias per negative column)
SM2_P = 2 ** 256 - 2 ** 224 - 2 ** 96 + 2 ** 64 - 1
P192_P = 2 ** 192 - 2 ** 64 - 1
SPARSE = {
    # u p = -u + u 2^64 - u 2^96 + u 2^224 (2^32 - 1)
    "SM2U": (SM2_P, 10, 28, False, [(2, 1 << 8), (3, -(1 << 12)), (8, (1 << 32) - 1)], 1 << 44),
    # u p = +u - u 2^96 + u 2^224            (p = 1 mod 2^27: u = -c_i)
    "P224U": (2 ** 224 - 2 ** 96 + 1, 9, 27, True, [(3, -(1 << 15)), (8, 1 << 8)], 1 << 46),
    # u p = -u - u 2^64 + u 2^192
    "P192U": (P192_P, 8, 26, False, [(2, -(1 << 12)), (7, 1 << 10)], 1 << 42),
}


def sparse_bias(name):
    """bias per column (2 nl + 1 of them): b on every column that receives a negative term, plus the canonical limbs of
    k p - (those) so that the biases sum to a multiple of p"""
    p, nl, bits, p0_one, terms, b = SPARSE[name]
    neg = sorted({i + off for i in range(nl) for off, coef in terms if coef < 0})
    target = sum(b << (bits * j) for j in neg)
    k = -(-target // p)
    rest = to_limbs(k * p - target, nl, bits)
    bias = [0] * (2 * nl + 1)
    for j in neg:
        bias[j] += b
    for j in range(nl):
        bias[j] += rest[j]
    assert sum(v << (bits * j) for j, v in enumerate(bias)) == k * p and all(v < (1 << 64) for v in bias)
    for off, coef in terms:
        if coef < 0:
            assert ((1 << bits) - 1) * -coef < b, "bias too small for the term"
    return bias


def sparse_mont_mul(name, a, b, c_extra=None, hi_add=None):
    p, nl, bits, p0_one, terms, _ = SPARSE[name]
    mask = (1 << bits) - 1
    c = list(sparse_bias(name))
    for x, y in ((a, b),) + ((c_extra,) if c_extra is not None else ()):
        for i in range(nl):
            for j in range(nl):
                c[i + j] = chk64(c[i + j] + chk32(x[i]) * chk32(y[j]))
    if hi_add is not None:
        for k in range(nl):
            c[nl + k] = chk64(c[nl + k] + chk32(hi_add[k]))
    for i in range(nl):
        if p0_one:
            u = (-c[i]) & mask
            carry = chk64(c[i] + u) >> bits
        else:
            u = c[i] & mask
            carry = c[i] >> bits
        c[i + 1] = chk64(c[i + 1] + carry)
        for off, coef in terms:
            v = c[i + off] + u * coef
            assert v >= 0, "column went negative: the bias is too small"
            c[i + off] = chk64(v)
    r = [0] * nl
    v = c[nl]
    for k in range(nl - 1):
        r[k] = v & mask
        v = chk64(c[nl + 1 + k] + (v >> bits))
    r[nl - 1] = chk32(v)
    return r


BIGN_P = 2 ** 256 - 189                          # bignp256/src/arithmetic/field.rs:60-66


def bign_mont_mul(a, b, c_extra=None, hi_add=None):
    """bign-curve256v1 on 10 x 28 limbs (R = 2^280) with SIGNED column accumulators and the Montgomery rows in sparse form:
    u p = -189 u + u 2^256 with u = c_i / 189 mod 2^28 — -189 u clears the low 28 bits of column i, + 16 u goes into column
    i + 9 (2^256 = 2^4 2^252).  Two multiply-adds per row instead of ten; product limit 11 (10 x 11 x LB^2 < 2^63)."""
    mask = (1 << 28) - 1
    pinv = pow(189, -1, 1 << 28)                  # = -p^-1 mod 2^28
    c = [0] * 21
    for x, y in ((a, b),) + ((c_extra,) if c_extra is not None else ()):
        for i in range(10):
            for j in range(10):
                c[i + j] = chk_s64(c[i + j] + chk32(x[i]) * chk32(y[j]))
    if hi_add is not None:
        for k in range(10):
            c[10 + k] = chk_s64(c[10 + k] + chk32(hi_add[k]))
    for i in range(10):
        u = (c[i] * pinv) & mask
        t = chk_s64(c[i] - 189 * u)
        assert t & mask == 0
        c[i + 1] = chk_s64(c[i + 1] + (t >> 28))
        c[i + 9] = chk_s64(c[i + 9] + 16 * u)
    r = [0] * 10
    v = c[10]
    for k in range(9):
        r[k] = v & mask
        v = chk_s64(c[11 + k] + (v >> 28))
    r[9] = chk32(v)
    return r


def selftest(trials=300, seed=1):
    rng = random.Random(seed)
    # sm2 / p224 / p192 sparse rows with bias: adversarial magnitudes at each set's product limit, fused pairs, random values
    for name, (lb, top1, maxprod, maxmag) in {"SM2U": (P_LB, 32, 24, 15), "P224U": (Q_LB, 512, 30, 28), "P192U": ((1 << 26) + (1 << 17), 2048, 30, 28)}.items():
        pp, nl, bits = SPARSE[name][0], SPARSE[name][1], SPARSE[name][2]
        rinv_s = pow(1 << (nl * bits), -1, pp)
        pairs = [(ma, maxprod // ma) for ma in range(1, maxmag + 1) if maxprod // ma <= maxmag and maxprod // ma >= 1]
        for ma, mb in pairs:
            a = [ma * lb - 1] * (nl - 1) + [top1 * ma - 1]
            b = [mb * lb - 1] * (nl - 1) + [top1 * mb - 1]
            r = sparse_mont_mul(name, a, b)
            assert from_limbs(r, bits) % pp == from_limbs(a, bits) * from_limbs(b, bits) * rinv_s % pp, name
        a, b, x, y = ([m * lb - 1] * (nl - 1) + [top1 * m - 1] for m in (4, maxprod // 8, 3, maxprod // 8))
        r = sparse_mont_mul(name, a, b, (x, y))
        assert from_limbs(r, bits) % pp == (from_limbs(a, bits) * from_limbs(b, bits) + from_limbs(x, bits) * from_limbs(y, bits)) * rinv_s % pp
        for _ in range(trials):
            a = [rng.randrange(lb) for _ in range(nl - 1)] + [rng.randrange(top1)]
            b = [rng.randrange(lb) for _ in range(nl - 1)] + [rng.randrange(top1)]
            r = sparse_mont_mul(name, a, b)
            assert from_limbs(r, bits) % pp == from_limbs(a, bits) * from_limbs(b, bits) * rinv_s % pp
            assert from_limbs(r, bits) < 2 * pp and all(0 <= v < (1 << bits) for v in r[:nl - 1]), name
    # bign256: adversarial magnitudes at the product limit 11 (a single product 11 x 1, and the fused pair 5 x 1 + 6 x 1)
    rinv = pow(1 << 280, -1, BIGN_P)
    for ma, mb in ((11, 1), (1, 11), (3, 3), (2, 5)):
        a = [ma * P_LB - 1] * 9 + [32 * ma - 1]
        b = [mb * P_LB - 1] * 9 + [32 * mb - 1]
        r = bign_mont_mul(a, b)
        assert from_limbs(r, 28) % BIGN_P == from_limbs(a, 28) * from_limbs(b, 28) * rinv % BIGN_P
    a, b, x, y = ([m * P_LB - 1] * 9 + [32 * m - 1] for m in (5, 1, 6, 1))
    r = bign_mont_mul(a, b, (x, y))
    assert from_limbs(r, 28) % BIGN_P == (from_limbs(a, 28) * from_limbs(b, 28) + from_limbs(x, 28) * from_limbs(y, 28)) * rinv % BIGN_P
    for _ in range(trials):
        a = [rng.randrange(P_LB) for _ in range(9)] + [rng.randrange(32)]
        b = [rng.randrange(P_LB) for _ in range(9)] + [rng.randrange(32)]
        r = bign_mont_mul(a, b)
        assert from_limbs(r, 28) % BIGN_P == from_limbs(a, 28) * from_limbs(b, 28) * rinv % BIGN_P
        assert all(0 <= v < (1 << 28) for v in r[:9]) and 0 <= r[9] < 64
    # k256: adversarial maxima at the magnitude-product limit 7 (e.g. 7 x 1, 3 x 2) and random values
    for ma, mb in ((7, 1), (1, 7), (3, 2), (2, 3), (1, 1), (2, 2)):
        assert ma * mb <= 7
        a = [ma * K_LB - 1] * 9
        b = [mb * K_LB - 1] * 9
        r = k256_mul(a, b)
        assert from_limbs(r, K_B) % K_P == from_limbs(a, K_B) * from_limbs(b, K_B) % K_P
    for _ in range(trials):
        ma, mb = rng.choice([(1, 1), (2, 2), (3, 2), (7, 1), (2, 3)])
        a = [rng.randrange(ma * K_LB) for _ in range(9)]
        b = [rng.randrange(mb * K_LB) for _ in range(9)]
        r = k256_mul(a, b)
        assert from_limbs(r, K_B) % K_P == from_limbs(a, K_B) * from_limbs(b, K_B) % K_P
    # the fused a * b - c (Field::mul_sub / sqr_sub): adversarial products at the limit with the largest subtrahends, then random
    for m in range(1, 7):
        assert from_limbs(k256_zconst(m), K_B) % K_P == 0
    for (ma, mb), m in (((7, 1), 6), ((1, 7), 6), ((3, 2), 4), ((2, 2), 3), ((1, 1), 1)):
        a = [ma * K_LB - 1] * 9
        b = [mb * K_LB - 1] * 9
        for c in ([0] * 9, [m * K_LB - 1] * 9):
            r = k256_mul_sub(a, b, c, m)
            assert from_limbs(r, K_B) % K_P == (from_limbs(a, K_B) * from_limbs(b, K_B) - from_limbs(c, K_B)) % K_P
    for _ in range(trials):
        (ma, mb), m = rng.choice([((1, 1), 1), ((2, 2), 2), ((1, 4), 2), ((1, 1), 3), ((3, 2), 6)])
        a = [rng.randrange(ma * K_LB) for _ in range(9)]
        b = [rng.randrange(mb * K_LB) for _ in range(9)]
        c = [rng.randrange(m * K_LB) for _ in range(9)]
        r = k256_mul_sub(a, b, c, m)
        assert from_limbs(r, K_B) % K_P == (from_limbs(a, K_B) * from_limbs(b, K_B) - from_limbs(c, K_B)) % K_P
        n = k256_norm([rng.randrange(1 << 32) for _ in range(9)])
    # k_to_words without the leading carry passes, on magnitude-1 inputs: the extremes, the neighbourhood of every multiple of p
    # that fits (canonical and shifted limb splits), random limbs
    cases = [[K_LB - 1] * 9, [0] * 9, [K_MASK] * 9, [K_LB - 1] * 8 + [0]]
    for mult in range(34):
        for d in (-2, -1, 0, 1, 2):
            v = mult * K_P + d
            if v < 0:
                continue
            a = [(v >> (K_B * i)) & K_MASK for i in range(8)] + [v >> (K_B * 8)]
            if a[8] < K_LB:
                cases.append(a)
            for i in range(8):
                if a[i + 1] > 0 and a[i] + (1 << K_B) < K_LB:
                    b = list(a)
                    b[i + 1] -= 1
                    b[i] += 1 << K_B
                    cases.append(b)
    for _ in range(20 * trials):
        cases.append([rng.randrange(K_LB) for _ in range(9)])
        cases.append([rng.choice([0, 1, K_MASK, K_MASK + 1, K_LB - 1, rng.randrange(K_LB)]) for _ in range(9)])
    for a in cases:
        assert k256_to_words_m1(a) == from_limbs(a, K_B) % K_P
    # Field::mul_sub / sqr_sub on the Montgomery fields: the subtrahend's limbs ((multiple of p) - c, anything below 2^32) enter
    # the high columns before the rows — products at each set's limit with the largest addends, then random ones; the value must
    # be a b / R + addend and no accumulator may overflow (the chk64 / chk_s64 inside the routines)
    def fused(mul, pp, nl, bits, lb, top1, maxprod, maxmag, label):
        rinv_f = pow(1 << (nl * bits), -1, pp)
        pairs = [(ma, maxprod // ma) for ma in range(1, maxmag + 1) if 1 <= maxprod // ma <= maxmag]
        for ma, mb in pairs:
            a = [ma * lb - 1] * (nl - 1) + [top1 * ma - 1]
            b = [mb * lb - 1] * (nl - 1) + [top1 * mb - 1]
            for h in ([(1 << 32) - 1] * (nl - 1) + [1 << 20], [0] * nl):
                r = mul(a, b, h)
                want = (from_limbs(a, bits) * from_limbs(b, bits) * rinv_f + from_limbs(h, bits)) % pp
                assert from_limbs(r, bits) % pp == want, label
                assert all(x < (1 << bits) for x in r[:nl - 1]), label
        for _ in range(trials // 4):
            ma, mb = rng.choice(pairs)
            a = [rng.randrange(ma * lb) for _ in range(nl - 1)] + [rng.randrange(top1 * ma)]
            b = [rng.randrange(mb * lb) for _ in range(nl - 1)] + [rng.randrange(top1 * mb)]
            h = [rng.randrange(1 << 32) for _ in range(nl - 1)] + [rng.randrange(1 << 20)]
            r = mul(a, b, h)
            assert from_limbs(r, bits) % pp == (from_limbs(a, bits) * from_limbs(b, bits) * rinv_f + from_limbs(h, bits)) % pp, label
    fused(lambda a, b, h: p256_mont_mul(a, b, hi_add=h), P_P, 10, 28, P_LB, 32, 23, 15, "p256")
    fused(lambda a, b, h: p384_mont_mul(a, b, hi_add=h), Q_P, 15, 27, Q_LB, Q_TOP1, 30, 28, "p384")
    fused(lambda a, b, h: bign_mont_mul(a, b, hi_add=h), BIGN_P, 10, 28, P_LB, 32, 11, 11, "bign256")
    for name, (lb, top1, maxprod, maxmag) in {"SM2U": (P_LB, 32, 24, 15), "P224U": (Q_LB, 512, 30, 28), "P192U": ((1 << 26) + (1 << 17), 2048, 30, 28)}.items():
        fused(lambda a, b, h, name=name: sparse_mont_mul(name, a, b, hi_add=h), SPARSE[name][0], SPARSE[name][1], SPARSE[name][2], lb, top1,
              maxprod, maxmag, name)
    bp256 = 0xA9FB57DBA1EEA9BC3E660A909D838D726E3BF623D52620282013481D1F6E5377
    bp384 = 0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B412B1DA197FB71123ACD3A729901D1A71874700133107EC53
    fused(lambda a, b, h: umont_mul_general(a, b, 10, 28, to_limbs(bp256, 10, 28), hi_add=h), bp256, 10, 28, P_LB, 32, 22, 15, "bp256")
    fused(lambda a, b, h: umont_mul_general(a, b, 15, 27, to_limbs(bp384, 15, 27), hi_add=h), bp384, 15, 27, Q_LB, 71, 24, 28, "bp384")
    p521 = 2 ** 521 - 1
    fused(lambda a, b, h: umont_mul(a, b, 20, 27, to_limbs(p521, 20, 27), hi_add=h), p521, 20, 27, Q_LB, 512, 30, 28, "p521")
    # p256: limb-magnitude product limit 23 (single products and the fused pairs of the group law)
    rinv = pow(P_R, -1, P_P)
    for (ma, mb), (mc, md) in (((4, 5), (1, 3)), ((12, 1), (11, 1)), ((4, 1), (3, 3))):
        a, b, x, y = ([m * P_LB - 1] * 9 + [32 * m - 1] for m in (ma, mb, mc, md))
        r = p256_mont_mul(a, b, (x, y))
        want = (from_limbs(a, P_B) * from_limbs(b, P_B) + from_limbs(x, P_B) * from_limbs(y, P_B)) * rinv % P_P
        assert from_limbs(r, P_B) % P_P == want and from_limbs(r, P_B) < 2 * P_P
    for ma, mb in ((15, 1), (1, 15), (4, 5), (5, 4), (1, 1), (3, 7), (11, 2), (2, 11)):
        top_a, top_b = 32 * ma, 32 * mb
        a = [ma * P_LB - 1] * 9 + [top_a - 1]
        b = [mb * P_LB - 1] * 9 + [top_b - 1]
        r = p256_mont_mul(a, b)
        assert from_limbs(r, P_B) % P_P == from_limbs(a, P_B) * from_limbs(b, P_B) * rinv % P_P
        assert from_limbs(r, P_B) < 2 * P_P
    for _ in range(trials):
        ma, mb = rng.choice([(1, 1), (4, 5), (15, 1), (2, 11)])
        a = [rng.randrange(ma * P_LB) for _ in range(9)] + [rng.randrange(32 * ma)]
        b = [rng.randrange(mb * P_LB) for _ in range(9)] + [rng.randrange(32 * mb)]
        r = p256_mont_mul(a, b)
        assert from_limbs(r, P_B) % P_P == from_limbs(a, P_B) * from_limbs(b, P_B) * rinv % P_P
        assert from_limbs(r, P_B) < 2 * P_P
    # p384: limb-magnitude product limit 30 (signed sparse rows), single magnitudes <= 28; the dense generic routine
    # (unsigned columns, limit 60) is the cross-check
    qinv = pow(Q_R, -1, Q_P)
    for ma, mb in ((28, 1), (1, 28), (5, 6), (1, 1), (10, 3), (15, 2)):
        a = [ma * Q_LB - 1] * 14 + [Q_TOP1 * ma - 1]
        b = [mb * Q_LB - 1] * 14 + [Q_TOP1 * mb - 1]
        r = p384_mont_mul(a, b)
        assert r == umont_mul(a, b, Q_NL, Q_B, Q_LIMBS)
        assert from_limbs(r, Q_B) % Q_P == from_limbs(a, Q_B) * from_limbs(b, Q_B) * qinv % Q_P
        assert from_limbs(r, Q_B) < 2 * Q_P and all(x < (1 << 27) for x in r[:14]) and r[14] < Q_TOP1
    for (ma, mb), (mc, md) in (((4, 5), (1, 3)), ((5, 5), (5, 1)), ((28, 1), (2, 1))):
        a, b, x, y = ([m * Q_LB - 1] * 14 + [Q_TOP1 * m - 1] for m in (ma, mb, mc, md))
        r = p384_mont_mul(a, b, (x, y))
        want = (from_limbs(a, Q_B) * from_limbs(b, Q_B) + from_limbs(x, Q_B) * from_limbs(y, Q_B)) * qinv % Q_P
        assert from_limbs(r, Q_B) % Q_P == want and from_limbs(r, Q_B) < 2 * Q_P
    for _ in range(trials // 3):
        ma, mb = rng.choice([(1, 1), (5, 6), (28, 1), (2, 15)])
        a = [rng.randrange(ma * Q_LB) for _ in range(14)] + [rng.randrange(Q_TOP1 * ma)]
        b = [rng.randrange(mb * Q_LB) for _ in range(14)] + [rng.randrange(Q_TOP1 * mb)]
        r = p384_mont_mul(a, b)
        assert r == umont_mul(a, b, Q_NL, Q_B, Q_LIMBS)
        assert from_limbs(r, Q_B) % Q_P == from_limbs(a, Q_B) * from_limbs(b, Q_B) * qinv % Q_P
        assert from_limbs(r, Q_B) < 2 * Q_P
    # sm2 through the generic (dense-row) routine: the value the sparse biased rows below must reproduce modulo p
    s_p = 2 ** 256 - 2 ** 224 - 2 ** 96 + 2 ** 64 - 1
    s_limbs = to_limbs(s_p, 10, 28)
    sinv = pow(P_R, -1, s_p)
    for ma, mb in ((15, 1), (4, 6), (3, 8), (12, 2), (1, 1)):
        a = [ma * P_LB - 1] * 9 + [32 * ma - 1]
        b = [mb * P_LB - 1] * 9 + [32 * mb - 1]
        r = umont_mul(a, b, 10, 28, s_limbs)
        assert from_limbs(r, 28) % s_p == from_limbs(a, 28) * from_limbs(b, 28) * sinv % s_p
        assert from_limbs(r, 28) < 2 * s_p and all(x < (1 << 28) for x in r[:9]) and r[9] < 32
    # the generic routine agrees with the p256-specific model
    for _ in range(20):
        a = [rng.randrange(P_LB) for _ in range(9)] + [rng.randrange(32)]
        b = [rng.randrange(P_LB) for _ in range(9)] + [rng.randrange(32)]
        assert umont_mul(a, b, P_NL, P_B, P_LIMBS) == p256_mont_mul(a, b)
    for m in range(1, 8):
        limbs, k = sub_constant(K_P, 9, 29, m, K_LB, K_LB)
        assert all(m * K_LB <= x < (m + 1) * K_LB + (1 << 29) for x in limbs), (m, [hex(x) for x in limbs])
    for m in range(1, 14):
        limbs, k = sub_constant(P_P, 10, 28, m, P_LB, 32)
        assert all(m * P_LB <= x < m * P_LB + (1 << 28) for x in limbs[:9]) and 32 * m <= limbs[9] < 32 * (m + 1) + 16, (m, limbs[9], k)
    return True


if __name__ == "__main__":
    print("field model selftest:", selftest())
