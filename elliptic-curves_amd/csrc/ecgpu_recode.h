// ecgpu_recode.h — scalar recodings used by the kernels (host+device, see ecgpu_field.h).
//
//  * signed fixed windows of W bits, digits in (-2^(W-1), 2^(W-1)], least significant first —
//    for W = 4 this is the digit set of `Radix16Decomposition::new`
//    (primeorder/src/tables/radix16.rs:35-61) up to the tie-break at |d| = 8 (the reference
//    emits -8 and carries, we emit +8 ... see signed_radix16_msb for the bit-exact variant);
//  * `signed_radix16_msb`: the reference's exact radix-16 digits [-8, 7] obtained in O(1) per
//    digit from k' = k + 0x88..8 (adding 8 to every nibble performs the reference's
//    recentring carry chain `carry = (d+8)>>4` in one multi-limb addition), so the kernel can
//    walk the digits most-significant-first without storing them;
//  * k256 GLV split `decompose_scalar` (k256/src/arithmetic/mul/glv.rs:149-156,
//    scalar/wide64.rs:64-119) on 32-bit limbs.
#pragma once

#include "ecgpu_field.h"

namespace ecgpu {

// bits [pos, pos+w) of a little-endian limb array of NL limbs (w <= 16), zero past the end
template <int NL>
ECGPU_HD uint32_t get_bits(const uint32_t* k, int pos, int w) {
    int limb = pos >> 5, sh = pos & 31;
    if (limb >= NL) return 0;
    uint32_t lo = k[limb] >> sh;
    if (sh + w > 32 && limb + 1 < NL) lo |= k[limb + 1] << (32 - sh);
    return lo & ((1u << w) - 1);
}

// One step of the least-significant-first signed window recoding.
// in: raw window value (w bits) and carry (0/1); out: digit in (-2^(w-1), 2^(w-1)], new carry.
ECGPU_HD int signed_window_step(uint32_t raw, int w, uint32_t* carry) {
    uint32_t v = raw + *carry;
    uint32_t half = 1u << (w - 1);
    if (v > half) {
        *carry = 1;
        return (int)v - (int)(1u << w);
    }
    *carry = 0;
    return (int)v;
}

// Number of signed windows needed for a `bits`-bit scalar: the top window only absorbs what is
// left plus the carry.
ECGPU_HD int signed_window_count(int bits, int w) { return bits / w + 1; }

// ---- Pippenger bucket keys -----------------------------------------------------------------------
// A folded scalar of kbits = bits - 1 significant bits is cut into nwin = kbits/c + 1 windows.  Windows 0..nwin-2 carry signed c-bit
// digits d in (-2^(c-1), 2^(c-1)], bucket = |d| - 1, weight = bucket + 1.  The last window holds the
// remaining r = kbits % c bits plus the carry as an UNSIGNED digit in [0, 2^r].  It has only 2^r distinct
// values, so without care all n terms of that window would pile into 2^r buckets (before folding, 256-bit
// scalars at c = 16 put half of all terms into ONE bucket: measured 85 s instead of 30 ms).  Its terms are therefore spread over all 2^(c-1)
// buckets with the low `shift = c-1-r` bits of the term index as a sub-bucket:
//     bucket = ((d - 1) << shift) | (index mod 2^shift),   weight(bucket) = (bucket >> shift) + 1.
//
// Scalar folding: a scalar with its top bit set is replaced by n - k (< 2^(bits-1) for all three curves,
// whose orders exceed 2^(bits-1)) and the sign of every digit is flipped: k P = (n - k)(-P).  The digits
// then cover only kbits = bits - 1 bits, so for c = 16 there are exactly 16 full windows instead of 16 plus
// a carry-only 17th.
ECGPU_HD int msm_top_shift(int kbits, int c) { return c - 1 - kbits % c; }

// k <- n - k if the top bit of k is set; returns whether it did.  (k must be < n.)
template <int NL>
ECGPU_HD bool fold_scalar(uint32_t* k, const uint32_t* order) {
    bool high = (k[NL - 1] >> 31) != 0;
    uint32_t d[NL];
    mp_sub<NL>(d, order, k);
#pragma unroll
    for (int i = 0; i < NL; i++) k[i] = high ? d[i] : k[i];
    return high;
}

struct MsmDigit {
    uint32_t bucket;
    uint32_t neg;      // 1: subtract the point
    bool nonzero;
};

// k: sub-scalar of kbits significant bits in NL words (the folded scalar, kbits = 32 NL - 1; a GLV half, kbits = 128),
// flip: its sign flag.  The returned sign already includes the flip.  nwin = signed_window_count(kbits, c).
template <int NL>
ECGPU_HD MsmDigit msm_digit(const uint32_t* k, int w, int c, int nwin, uint32_t* carry, uint32_t term_index, bool flip,
                            int kbits = 32 * NL - 1) {
    MsmDigit r;
    r.bucket = 0; r.neg = 0; r.nonzero = false;
    if (w < nwin - 1) {
        int d = signed_window_step(get_bits<NL>(k, w * c, c), c, carry);
        if (d != 0) {
            r.nonzero = true;
            r.neg = (d < 0) != flip;
            r.bucket = (uint32_t)(d < 0 ? -d : d) - 1;
        }
    } else {
        const int rem = kbits % c;
        const int shift = c - 1 - rem;
        uint32_t d = (rem ? get_bits<NL>(k, w * c, rem) : 0u) + *carry;
        *carry = 0;
        if (d != 0) {
            r.nonzero = true;
            r.neg = flip;
            r.bucket = ((d - 1) << shift) | (term_index & ((1u << shift) - 1));
        }
    }
    return r;
}

// The same digits in window order (w = 0, 1, ..) WITHOUT a dynamically indexed scalar word: the sub-scalar sits in kw[] and moves
// down one word whenever 32 bits have been consumed, so that every window is one funnel shift of (kw[1] : kw[0]) by an offset
// below 32.  msm_digit(k, w, ..) extracts bits [w c, w c + c) of a register array at a run-time position — a chain of selects per
// word and half of k_msm_prepare's vector instructions; here a digit costs one v_alignbit, a mask and the recoding step, and one
// wave-uniform branch (the offset depends on w and c only) moves the words every 32 / c digits.
template <int NL>
struct MsmDigitStream {
    static_assert(NL >= 2, "at least two words");
    uint32_t kw[NL + 1];
    uint32_t off, carry;
    ECGPU_HD void init(const uint32_t* k) {
#pragma unroll
        for (int i = 0; i < NL; i++) kw[i] = k[i];
        kw[NL] = 0;
        off = 0;
        carry = 0;
    }
    // the next `bits` (0 .. 16) bits of the scalar, zero past its end
    ECGPU_HD uint32_t take(int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t v = __builtin_amdgcn_alignbit(kw[1], kw[0], off) & ((1u << bits) - 1);
#else
        const uint32_t v = (uint32_t)((((uint64_t)kw[1] << 32) | kw[0]) >> off) & ((1u << bits) - 1);
#endif
        off += (uint32_t)bits;
        if (off >= 32) {
            off -= 32;
#pragma unroll
            for (int i = 0; i < NL; i++) kw[i] = kw[i + 1];
        }
        return v;
    }
    // digit of window w; calls must come with w = 0, 1, .., nwin - 1 (arguments as for msm_digit)
    ECGPU_HD MsmDigit next(int w, int c, int nwin, uint32_t term_index, bool flip, int kbits = 32 * NL - 1) {
        MsmDigit r;
        r.bucket = 0; r.neg = 0; r.nonzero = false;
        if (w < nwin - 1) {
            int d = signed_window_step(take(c), c, &carry);
            if (d != 0) {
                r.nonzero = true;
                r.neg = (d < 0) != flip;
                r.bucket = (uint32_t)(d < 0 ? -d : d) - 1;
            }
        } else {
            const int rem = kbits % c;
            const int shift = c - 1 - rem;
            uint32_t d = take(rem) + carry;
            carry = 0;
            if (d != 0) {
                r.nonzero = true;
                r.neg = flip;
                r.bucket = ((d - 1) << shift) | (term_index & ((1u << shift) - 1));
            }
        }
        return r;
    }
};

// k' = k + 0x8888...8 over NL limbs (+1 limb for the carry). digit(i) = nibble_i(k') - 8 for
// i < 8*NL, and nibble (0/1) for i = 8*NL: exactly Radix16Decomposition::new's output.
template <int NL>
struct Radix16Msb {
    uint32_t kp[NL + 1];
    ECGPU_HD void init(const uint32_t* k) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            c += (uint64_t)k[i] + 0x88888888u;
            kp[i] = (uint32_t)c;
            c >>= 32;
        }
        kp[NL] = (uint32_t)c;
    }
    ECGPU_HD int digit(int i) const {
        uint32_t nib = (kp[i >> 3] >> ((i & 7) * 4)) & 0xF;
        return (i < 8 * NL) ? (int)nib - 8 : (int)nib;
    }
};

// The same trick for windows of W bits: k' = k + sum_{j < FULL} 2^(W j + W - 1) recentres every full window at once, digit j =
// window_j(k') - 2^(W-1) in [-2^(W-1), 2^(W-1) - 1] for j < FULL = BITS / W, and window FULL holds what is left of k' (the top
// BITS mod W bits plus the carry: at most 2^(W-1)) as an unsigned digit.  No step depends on the value of a digit.
template <int NL, int W, int BITS>
struct SignedWindowsMsb {
    ECGPU_CONST int FULL = BITS / W, COUNT = FULL + 1, HALF = 1 << (W - 1);
    static_assert(BITS <= 32 * NL && W >= 2 && W <= 16, "window geometry");
    struct HalfWords {
        uint32_t w[NL + 1];
        constexpr HalfWords() : w{} {
            for (int j = 0; j < FULL; j++) w[(W * j + W - 1) / 32] |= 1u << ((W * j + W - 1) % 32);
        }
    };
    ECGPU_CONST HalfWords HW{};
    uint32_t kp[NL + 1];
    ECGPU_HD void init(const uint32_t* k) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            c += (uint64_t)k[i] + HW.w[i];
            kp[i] = (uint32_t)c;
            c >>= 32;
        }
        kp[NL] = (uint32_t)c;
    }
    // i = 0 .. FULL (a public loop counter)
    ECGPU_HD int digit(int i) const {
        const uint32_t raw = get_bits<NL + 1>(kp, W * i, W);
        return i < FULL ? (int)raw - HALF : (int)raw;
    }
};

// ---------------------------------------------------------------------------------------------
// secp256k1 scalar arithmetic mod n on 8x32 limbs + GLV decomposition
// ---------------------------------------------------------------------------------------------

struct K256Scalar {
    using C = K256Params;
    // 2^256 - n  (129 bits)                      k256 scalar/wide64.rs:11 NEG_MODULUS
    ECGPU_CONST uint32_t NEG_N[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u};
    // (n - 1) / 2                                 k256 scalar.rs FRAC_MODULUS_2
    ECGPU_CONST uint32_t HALF_N[8] = {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u,
                                      0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
    // GLV constants, little-endian limbs           k256 mul/glv.rs:10-37
    ECGPU_CONST uint32_t MINUS_LAMBDA[8] = {0xB51283CFu, 0xE0CFC810u, 0x8EC739C2u, 0xA880B9FCu,
                                            0x77ED9BA4u, 0x5AD9E3FDu, 0x3FA3CF1Fu, 0xAC9C52B3u};
    ECGPU_CONST uint32_t MINUS_B1[8] = {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u,
                                        0, 0, 0, 0};
    ECGPU_CONST uint32_t MINUS_B2[8] = {0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u,
                                        0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    ECGPU_CONST uint32_t G1[8] = {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u,
                                  0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u};
    ECGPU_CONST uint32_t G2[8] = {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu,
                                  0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u};

    // r = a + b mod n (inputs < n)                 k256 scalar.rs:106-108
    static ECGPU_HD void add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
        uint32_t t[8], d[8];
        uint32_t carry = mp_add<8>(t, a, b);
        uint32_t borrow = mp_sub<8>(d, t, C::ORDER);
        bool use_d = carry || !borrow;
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = use_d ? d[i] : t[i];
    }
    static ECGPU_HD void neg(uint32_t* r, const uint32_t* a) {           // scalar.rs:100-102
        uint32_t d[8];
        bool z = mp_is_zero<8>(a);
        mp_sub<8>(d, C::ORDER, a);
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = z ? 0u : d[i];
    }
    static ECGPU_HD bool is_high(const uint32_t* a) {                    // scalar.rs:419-423
        uint32_t t[8];
        return mp_sub<8>(t, HALF_N, a) != 0;  // HALF_N < a
    }

    // acc[0..na+5) = x[0..na) * NEG_N, helper for the folding reduction
    template <int NA>
    static ECGPU_HD void mul_neg_n(uint32_t* out, const uint32_t* x) {
#pragma unroll
        for (int i = 0; i < NA + 5; i++) out[i] = 0;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            uint32_t carry = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                uint64_t t = (uint64_t)x[i] * NEG_N[j] + out[i + j] + carry;
                out[i + j] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
            out[i + 5] = carry;
        }
    }

    // 512-bit l -> l mod n by folding the high part with 2^256 = NEG_N (mod n) three times and one
    // conditional subtraction                      k256 scalar/wide64.rs:121-212
    static ECGPU_HD void reduce_wide(uint32_t* r, const uint32_t* l) {
        // fold 1: m = l[0..8) + l[8..16) * NEG_N        (< 2^386, 13 limbs)
        uint32_t m[13], prod[13];
        mul_neg_n<8>(prod, l + 8);
        {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 13; i++) {
                c += (uint64_t)prod[i] + (i < 8 ? l[i] : 0u);
                m[i] = (uint32_t)c;
                c >>= 32;
            }
        }
        // fold 2: p = m[0..8) + m[8..13) * NEG_N        (< 2^259, 10 limbs used)
        uint32_t p[10], prod2[10];
        mul_neg_n<5>(prod2, m + 8);
        {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 10; i++) {
                c += (uint64_t)prod2[i] + (i < 8 ? m[i] : 0u);
                p[i] = (uint32_t)c;
                c >>= 32;
            }
        }
        // fold 3: t = p[0..8) + p[8] * NEG_N            (p[9] == 0, p[8] <= 4)
        uint32_t t[9];
        {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                c += (uint64_t)p[i] + (i < 5 ? (uint64_t)p[8] * NEG_N[i] : 0u);
                t[i] = (uint32_t)c;
                c >>= 32;
            }
            t[8] = (uint32_t)c;
        }
        uint32_t d[8];
        uint32_t borrow = mp_sub<8>(d, t, C::ORDER);
        bool use_d = t[8] || !borrow;
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = use_d ? d[i] : t[i];
    }

    static ECGPU_HD void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {  // scalar.rs:120-122
        uint32_t l[16];
        mp_mul<8>(l, a, b);
        reduce_wide(r, l);
    }

    // round(a*b / 2^384)                            wide64.rs:64-119 with shift = 384
    static ECGPU_HD void mul_shift_384(uint32_t* r, const uint32_t* a, const uint32_t* b) {
        uint32_t l[16];
        mp_mul<8>(l, a, b);
        uint32_t res[8] = {l[12], l[13], l[14], l[15], 0, 0, 0, 0};
        uint32_t round_bit = l[11] >> 31;
        uint32_t one[8] = {round_bit, 0, 0, 0, 0, 0, 0, 0};
        add(r, res, one);
    }

    // ---- the GLV split on 29-bit limbs -----------------------------------------------------------------------------------------
    // k -> (r1, r2) with r1 + r2 lambda = k (mod n)    mul/glv.rs:149-156
    // The reference computes c1 = round(k g1 / 2^384) (-b1), c2 = round(k g2 / 2^384) (-b2), r2 = c1 + c2, r1 = k - r2 lambda,
    // all modulo n.  As INTEGERS, with the lattice vectors (a1, b1), (a2, b2) (a_i + b_i lambda = 0 mod n; a1 = b2):
    //      r2 = t1 |b1| - t2 b2,      r1 = k - t1 a1 - t2 a2,      |r1|, |r2| < 2^128
    // — the same residues, so both only need to be computed modulo 2^145 (five 29-bit limbs), sign = bit 144.  Everything
    // runs on 29-bit limbs with 64-bit column accumulators that cannot overflow (the field's own technique, ecgpu_field.h):
    // pure v_mad_u64_u32 chains without carries between the products.  The first version (32-bit words, operand scanning with a
    // carry per product, a 256 x 256-bit product + reduction mod n for r2 lambda) compiled to 1,840 instructions, 900 of them
    // register moves, and was a third of k_msm_prepare in GLV mode; this one is ~500.
    ECGPU_CONST uint32_t L29 = (1u << 29) - 1;
    ECGPU_CONST uint32_t G1_L29[9] = {0x05DBB031u, 0x049904D2u, 0x1A329FFAu, 0x151428E3u, 0x0EB153DAu, 0x08724942u, 0x0F37A1B2u, 0x0434FA8Du, 0x003086D2u};
    ECGPU_CONST uint32_t G2_L29[9] = {0x0AC47F71u, 0x0B8DA574u, 0x1D41B185u, 0x0411593Bu, 0x1E4C4221u, 0x1FD4855Fu, 0x00A1BD51u, 0x1AC021D1u, 0x00E4437Eu};
    ECGPU_CONST uint32_t MINUS_B1_L29[5] = {0x0ABFE4C3u, 0x1AA3FD48u, 0x03A20A1Bu, 0x06FDAC02u, 0x00000E44u};   // |b1|
    ECGPU_CONST uint32_t B2_L29[5] = {0x1284EB15u, 0x03648724u, 0x151AF37Au, 0x0DA4434Fu, 0x00000308u};         // b2 = a1
    ECGPU_CONST uint32_t A2_L29[5] = {0x1D44CFD8u, 0x1E08846Cu, 0x18BCFD95u, 0x14A1EF51u, 0x0000114Cu};         // a2 (129 bits)

    // t = round(k g / 2^384) as five 29-bit limbs (t <= 2^128); kl: k on nine 29-bit limbs, gl: g likewise
    static ECGPU_HD void mul_shift_384_l29(uint32_t* t, const uint32_t* kl, const uint32_t* gl) {
        uint64_t c[18];
#pragma unroll
        for (int i = 0; i < 18; i++) c[i] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
#pragma unroll
            for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)kl[i] * gl[j];         // < 9 * 2^58 per column
        }
        // exact base-2^29 digits 13 .. 17 of the product (bit 384 = bit 7 of digit 13), the ones below only through their carry
        uint64_t v = c[0];
#pragma unroll
        for (int i = 1; i <= 13; i++) v = c[i] + (v >> 29);
        uint32_t d[6];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            d[i] = (uint32_t)v & L29;
            v = c[14 + i] + (v >> 29);
        }
        d[4] = (uint32_t)v & L29;                 // digit 17 (< 2^19: the product is below 2^512)
        d[5] = 0;
        uint32_t carry = (d[0] >> 6) & 1u;        // the rounding bit, bit 383
#pragma unroll
        for (int j = 0; j < 5; j++) {
            uint32_t x = (((d[j] >> 7) | (d[j + 1] << 22)) & L29) + carry;
            t[j] = x & L29;
            carry = x >> 29;
        }
    }
    // five signed 64-bit columns -> magnitude (four 32-bit words) and sign of the value they hold modulo 2^145
    static ECGPU_HD bool signed_columns_to_words(uint32_t* m, const int64_t* col) {
        uint32_t l[5];
        int64_t v = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            v += col[i];
            l[i] = (uint32_t)v & L29;
            v >>= 29;                              // arithmetic: the borrow travels as -1
        }
        const bool negative = (l[4] >> 28) != 0;   // bit 144
        {                                          // two's complement over 145 bits if negative
            const uint32_t x = negative ? L29 : 0u;
            uint32_t carry = negative ? 1u : 0u;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                uint32_t y = (l[i] ^ x) + carry;
                l[i] = y & L29;
                carry = y >> 29;
            }
        }
        m[0] = l[0] | (l[1] << 29);
        m[1] = (l[1] >> 3) | (l[2] << 26);
        m[2] = (l[2] >> 6) | (l[3] << 23);
        m[3] = (l[3] >> 9) | (l[4] << 20);          // |value| < 2^128: l[4] < 2^12
        return negative;
    }
    // r1 = (neg1 ? -m1 : m1), r2 = (neg2 ? -m2 : m2), magnitudes below 2^128 as four words each
    static ECGPU_HD void decompose_signed(uint32_t* m1, bool* neg1, uint32_t* m2, bool* neg2, const uint32_t* k) {
        uint32_t kl[9];
#pragma unroll
        for (int l = 0; l < 9; l++) {
            const int bit = 29 * l, i = bit / 32, sh = bit % 32;
            uint64_t x = (uint64_t)k[i] >> sh;
            if (i + 1 < 8) x |= (uint64_t)k[i + 1] << (32 - sh);
            kl[l] = (uint32_t)x & L29;
        }
        uint32_t t1[5], t2[5];
        mul_shift_384_l29(t1, kl, G1_L29);
        mul_shift_384_l29(t2, kl, G2_L29);
        // columns 0 .. 4 only (everything is taken modulo 2^145); three unsigned accumulations, < 10 * 2^58 per column
        uint64_t u1[5], up[5], uq[5];
#pragma unroll
        for (int i = 0; i < 5; i++) u1[i] = up[i] = uq[i] = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int j = 0; i + j < 5; j++) {
                u1[i + j] += (uint64_t)t1[i] * B2_L29[j];            // t1 a1 + t2 a2
                u1[i + j] += (uint64_t)t2[i] * A2_L29[j];
                up[i + j] += (uint64_t)t1[i] * MINUS_B1_L29[j];      // t1 |b1|
                uq[i + j] += (uint64_t)t2[i] * B2_L29[j];            // t2 b2
            }
        }
        int64_t c1[5], c2[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            c1[i] = (int64_t)kl[i] - (int64_t)u1[i];
            c2[i] = (int64_t)up[i] - (int64_t)uq[i];
        }
        *neg1 = signed_columns_to_words(m1, c1);
        *neg2 = signed_columns_to_words(m2, c2);
    }
    // the same split as canonical residues modulo n (what the reference returns)
    static ECGPU_HD void decompose(uint32_t* r1, uint32_t* r2, const uint32_t* k) {
        uint32_t m1[8], m2[8];
        bool n1, n2;
        decompose_signed(m1, &n1, m2, &n2, k);
#pragma unroll
        for (int i = 4; i < 8; i++) m1[i] = m2[i] = 0;
        uint32_t d1[8], d2[8];
        mp_sub<8>(d1, C::ORDER, m1);
        mp_sub<8>(d2, C::ORDER, m2);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            r1[i] = n1 ? d1[i] : m1[i];
            r2[i] = n2 ? d2[i] : m2[i];
        }
    }
};

// ---- how an MSM term's scalar is cut into sub-scalars before the bucket windows ---------------------------------------
// Plain (every curve): one sub-term per term, the folded scalar (k -> n - k when the top bit is set, sign flag): 32 N - 1 bits.
// GLV (k256 only): the two halves k = r1 + r2 lambda (mod n), |r_i| < 2^128 after folding their signs into flags
// (k256/src/arithmetic/mul.rs:112-132, mul/glv.rs:149-156), the second one against lambda P = (beta x, y): 2 n sub-terms of
// 128 bits.  The additions are about the same in number (2 n x 8.5 windows against n x 16), but there are 9 windows of
// buckets to reduce instead of 16 and the Horner chain over the window sums needs 128 doublings instead of 240 — the part
// of an MSM that does not shrink with n.  It costs a decomposition per term and doubles the array the point gathers
// range over, so it pays for MSMs of up to a few million terms (a GPU's share of a sharded 2^24-term MSM), not beyond.
template <class C, bool GLV>
struct MsmSplit {
    static_assert(!GLV, "the GLV split exists for secp256k1 only");
    static constexpr int SUB = 1;
    static constexpr int KW = C::N;
    static constexpr int KBITS = 32 * C::N - 1;
    static ECGPU_HD void split(const uint32_t* k, uint32_t (*sub)[KW], bool* neg) {
#pragma unroll
        for (int i = 0; i < KW; i++) sub[0][i] = k[i];
        neg[0] = fold_scalar<KW>(sub[0], C::ORDER);
    }
};
template <>
struct MsmSplit<K256Params, true> {
    static constexpr int SUB = 2;
    static constexpr int KW = 4;
    static constexpr int KBITS = 128;
    static ECGPU_HD void split(const uint32_t* k, uint32_t (*sub)[KW], bool* neg) {
        K256Scalar::decompose_signed(sub[0], &neg[0], sub[1], &neg[1], k);
    }
};
template <class C>
struct MsmHasGlv { static constexpr bool value = false; };
template <>
struct MsmHasGlv<K256Params> { static constexpr bool value = true; };

}  // namespace ecgpu
