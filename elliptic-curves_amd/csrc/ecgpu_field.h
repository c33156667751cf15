// ecgpu_field.h — prime-field arithmetic for the MI355X scalar-mul engine.
//
// Every function is `__host__ __device__`: the very same code the gfx950 kernels run is compiled with
// g++ and checked on a CPU against the oracle (tests/hostcheck/).
//
// Why unsaturated limbs.  Measured on MI355X (profiles/r01/isa_issue_rates.txt): v_mad_u64_u32
// (32x32+64 -> 64) issues in 4 cycles per wave64 — the same as a 64-bit add (v_lshl_add_u64) or any
// other VOP3 instruction — and plain 32-bit VOP2 ops (v_add_u32, v_and_b32, v_mov_b32) in 2.  The
// first version of this file used 8 saturated 32-bit limbs; its multiplication spent 74 MADs but 300
// further instructions on zero-extensions and 64-bit carry adds, and every field addition was a carry
// chain.  With limbs of 29 (k256) / 28 (p256) bits the 64-bit column accumulators cannot overflow, so a
// product is nothing but multiply-adds, and additions/subtractions are independent 32-bit ops per limb
// with no carries at all ("lazy" reduction — the same idea as the reference's own 5x52 / 10x26 fields,
// k256/src/arithmetic/field/field_5x52.rs:203-236, field_10x26.rs:258-290).
//
// Representations (C::REPR):
//   REPR_U29_K256  k256: 9 limbs x 29 bits, plain residues, value < M * 2^261; products are folded with
//                  2^261 = 256 * 2^29 + 31264 (mod p)
//   REPR_U28_MONT  p256: 10 limbs x 28 bits, Montgomery form R = 2^280 (p = -1 mod 2^28, so p' = 1 as in
//                  p256/src/arithmetic/field/field64.rs:59), value < M * 2p; p384: 15 limbs x 27 bits, R = 2^405
//                  (the reference: p384/src/arithmetic/field.rs:52-57 -> crypto-bigint ConstMontyForm)
//
// Lazy reduction needs bounds.  Elements carry their bounds in the type: Mag<C, L, V> has limbs
// <= L * LB ("limb magnitude") and value <= V * (value unit) ("value magnitude"); add/sub/neg grow the
// magnitudes, mul/sqr require L_a * L_b <= MAXPROD and return magnitude (1, 1), norm() resets the limb
// magnitude.  Every precondition is a static_assert, i.e. the bound analysis of tools/field_model.py is
// re-checked by the compiler at each call site — the compile-time twin of the reference's debug-build
// magnitude checker (k256/src/arithmetic/field/field_impl.rs:17-22).
#pragma once

#include "ecgpu_params.h"
#include "ecgpu_modinv.h"

namespace ecgpu {

// -DECGPU_FUSED_SUB=0 builds Field::mul_sub / sqr_sub as norm(sub(mul(..), ..)) — the form the point formulas had before the
// differences moved into the reductions (A/B measurements: profiles/r04/k256_fused_sub_ab.txt, mont_fused_sub_ab.txt)
#ifndef ECGPU_FUSED_SUB
#define ECGPU_FUSED_SUB 1
#endif

template <class C, int L, int V>
struct Mag {
    Fe<C::NL> e;
};

// A compile-time constant the optimiser must treat as unknown (device only).  v_mad_u64_u32 issues at the same
// rate as ANY 64-bit VALU op, so multiplying a limb by 2^k or 2^k - 1 and accumulating is one instruction as a
// multiply-add but three to five (moves, 64-bit shift, 64-bit add/sub) once the compiler strength-reduces it.
// Routing such constants through an SGPR the compiler cannot see into keeps them as multiply-adds.
static ECGPU_HD uint32_t opaque_const(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(x));
#endif
    return x;
}

template <class C>
struct Field {
    ECGPU_CONST int N = C::N;     // canonical 32-bit words
    ECGPU_CONST int NL = C::NL;   // limbs in registers
    ECGPU_CONST int NS = (C::NL / 4 + 1) * 4;   // words of the raw (lazy) storage form: a 16-byte multiple with at least one spare word
    ECGPU_CONST int REPR = C::REPR;
    using E = Fe<C::NL>;
    using M1 = Mag<C, 1, 1>;

    // largest allowed sum of limb-magnitude products in one column accumulation, largest limb magnitude
    ECGPU_CONST int MAXPROD = REPR == REPR_U29_K256 ? 7 : C::UC::MAXPROD;
    ECGPU_CONST int MAXMAG = REPR == REPR_U29_K256 ? 7 : C::UC::MAXMAG;

    // =============================================================================================
    // k256, 9 x 29  (model: tools/field_model.py k256_*)
    // =============================================================================================
    using KC = consts::K256U;
    ECGPU_CONST uint32_t KMASK = (1u << 29) - 1;

    // fold a 64-bit column of weight 2^(261 + 29 j) into columns j, j+1, j+2 through its 32-bit halves:
    // 2^261 = F1*2^29 + F0 and 2^(261+32) = G2*2^58 + G1*2^29 (mod p)
    static ECGPU_HD void k_fold(uint64_t* lo, int j, uint64_t col) {
        uint32_t cl = (uint32_t)col, ch = (uint32_t)(col >> 32);
        lo[j] += (uint64_t)cl * KC::F0;
        lo[j + 1] += (uint64_t)cl * opaque_const(KC::F1);
        lo[j + 1] += (uint64_t)ch * KC::G1;
        lo[j + 2] += (uint64_t)ch * opaque_const(KC::G2);
    }
    // 17 product columns -> 9 limbs of magnitude 1
    static ECGPU_HD E k_reduce(uint64_t* c) {
        uint64_t lo[11];
#pragma unroll
        for (int k = 0; k < 9; k++) lo[k] = c[k];
        lo[9] = 0;
        lo[10] = 0;
        // High columns 9..15: the upper 32 bits of a column move into the NEXT high column first (2^32 = 8 * 2^29, one
        // multiply-add by 8 — a 64-bit shift + mask + add would cost three issue slots: c[k + 1] < 56 LB^2 + 2^35 < 2^64), so that only the lower half is left to fold — two multiply-adds
        // instead of four per column.  The last one (and the second-stage column 10) has no next column and folds in full.
#pragma unroll
        for (int k = 9; k < 16; k++) {
            const uint32_t cl = (uint32_t)c[k], ch = (uint32_t)(c[k] >> 32);
            c[k + 1] += (uint64_t)ch * opaque_const(8u);
            lo[k - 9] += (uint64_t)cl * KC::F0;
            lo[k - 8] += (uint64_t)cl * opaque_const(KC::F1);
        }
        k_fold(lo, 7, c[16]);
        {
            uint64_t c9 = lo[9], c10 = lo[10];
            const uint32_t cl = (uint32_t)c9, ch = (uint32_t)(c9 >> 32);
            c10 += (uint64_t)ch * opaque_const(8u);
            lo[0] += (uint64_t)cl * KC::F0;
            lo[1] += (uint64_t)cl * opaque_const(KC::F1);
            k_fold(lo, 1, c10);
        }
        E r;
        uint64_t v = lo[0];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            r.v[k] = (uint32_t)v & KMASK;
            v = lo[k + 1] + (v >> 29);
        }
        r.v[8] = (uint32_t)v & KMASK;
        uint64_t top = v >> 29;                       // weight 2^261, < 2^36
        uint32_t tl = (uint32_t)top, th = (uint32_t)(top >> 32);
        uint64_t t0 = (uint64_t)tl * KC::F0 + r.v[0];
        uint64_t t1 = (uint64_t)tl * opaque_const(KC::F1) + r.v[1];
        t1 += (uint64_t)th * KC::G1;
        uint64_t t2 = (uint64_t)th * opaque_const(KC::G2) + r.v[2];
        r.v[0] = (uint32_t)t0 & KMASK;
        t1 += t0 >> 29;
        r.v[1] = (uint32_t)t1 & KMASK;
        t2 += t1 >> 29;
        r.v[2] = (uint32_t)t2 & KMASK;
        r.v[3] += (uint32_t)(t2 >> 29);               // slack absorbed by LB
        return r;
    }
    static ECGPU_HD void k_columns(uint64_t* c, const uint32_t* a, const uint32_t* b, bool accumulate) {
        if (!accumulate) {
#pragma unroll
            for (int k = 0; k < 17; k++) c[k] = 0;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
#pragma unroll
            for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a[i] * b[j];
        }
    }
    static ECGPU_HD void k_columns_sqr(uint64_t* c, const uint32_t* a) {
        uint32_t a2[9];
#pragma unroll
        for (int k = 0; k < 17; k++) c[k] = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) a2[j] = a[j] << 1;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            c[2 * i] += (uint64_t)a[i] * a[i];
#pragma unroll
            for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a[i] * a2[j];
        }
    }
    // carry-propagate a lazy element (limbs < 2^32) and fold the top: magnitude 1
    static ECGPU_HD E k_norm(const E& a) {
        E r;
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            uint32_t v = a.v[k] + carry;
            r.v[k] = v & KMASK;
            carry = v >> 29;
        }
        r.v[0] += carry * KC::F0;
        r.v[1] += carry * KC::F1;
        return r;
    }
    // a * k for a small constant, normalised
    static ECGPU_HD E k_mul_small(const E& a, uint32_t k) {
        E r;
        uint64_t t = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            t += (uint64_t)a.v[i] * k;
            r.v[i] = (uint32_t)t & KMASK;
            t >>= 29;
        }
        uint32_t top = (uint32_t)t;                    // < 2^32 / 2^29 * k
        uint64_t t0 = (uint64_t)top * KC::F0 + r.v[0];
        uint64_t t1 = (uint64_t)top * KC::F1 + r.v[1] + (t0 >> 29);
        r.v[0] = (uint32_t)t0 & KMASK;
        r.v[1] = (uint32_t)t1 & KMASK;
        r.v[2] += (uint32_t)(t1 >> 29);
        return r;
    }
    // exact canonical value in [0, p) as 8 little-endian 32-bit words.  NORMED: the limbs are already below LB (a magnitude-1
    // element, e.g. a fresh product) — the two carry passes that bring a lazy element there are skipped (the two folds below
    // start from limbs < 2^32 either way; model with adversarial inputs: tools/field_model.py k256_to_words_m1)
    template <bool NORMED = false>
    static ECGPU_HD void k_to_words(uint32_t* w, const E& a) {
        E r;
        if constexpr (NORMED) r = a;
        else r = k_norm(k_norm(a));                   // limbs < 2^29 + tiny, value < 2^261 + tiny
        // fold everything at or above 2^256 with 2^256 = 2^32 + 977 = 8 * 2^29 + 977, twice, exactly
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t v = r.v[k] + carry;
                r.v[k] = v & KMASK;
                carry = v >> 29;
            }
            uint32_t v8 = r.v[8] + carry;
            uint32_t e = v8 >> 24;
            r.v[8] = v8 & 0xFFFFFFu;
            r.v[0] += e * 977u;
            r.v[1] += e * 8u;
        }
        {
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t v = r.v[k] + carry;
                r.v[k] = v & KMASK;
                carry = v >> 29;
            }
            r.v[8] += carry;                           // value < 2^256 + 2^40: r.v[8] <= 2^24
        }
        // r >= p  <=>  r + (2^256 - p) >= 2^256, with 2^256 - p = 2^32 + 977
        E s = r;
        s.v[0] += 977u;
        s.v[1] += 8u;
        {
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t v = s.v[k] + carry;
                s.v[k] = v & KMASK;
                carry = v >> 29;
            }
            s.v[8] += carry;
        }
        bool ge = (s.v[8] >> 24) != 0;
        s.v[8] &= 0xFFFFFFu;
#pragma unroll
        for (int k = 0; k < 9; k++) r.v[k] = ge ? s.v[k] : r.v[k];
        // 9 x 29 -> 8 x 32
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int bit = 32 * i, l = bit / 29, sh = bit % 29;
            uint64_t x = (uint64_t)r.v[l] >> sh;
            x |= (uint64_t)r.v[l + 1] << (29 - sh);
            if (l + 2 < 9) x |= (uint64_t)r.v[l + 2] << (58 - sh);
            w[i] = (uint32_t)x;
        }
    }
    template <int NLIMB, int BITS>
    static ECGPU_HD E words_to_limbs(const uint32_t* w) {   // N words -> NLIMB limbs of BITS bits
        E r;
#pragma unroll
        for (int l = 0; l < NLIMB; l++) {
            int bit = BITS * l, i = bit / 32, sh = bit % 32;
            uint64_t x = i < N ? (uint64_t)w[i] >> sh : 0;
            if (i + 1 < N) x |= (uint64_t)w[i + 1] << (32 - sh);
            r.v[l] = (uint32_t)x & ((1u << BITS) - 1);
        }
        return r;
    }

    // =============================================================================================
    // unsaturated Montgomery: p256 10 x 28 (R = 2^280), p384 15 x 27 (R = 2^405); p = -1 mod 2^B so p' = 1
    // (models: tools/field_model.py p256_mont_mul / umont_mul)
    // =============================================================================================
    using PC = typename C::UC;
    ECGPU_CONST int UN = PC::NL, UB = PC::B;
    ECGPU_CONST uint32_t PMASK = (1u << PC::B) - 1;

    // 2*UN-1 product columns (c[2*UN-1], c[2*UN] zero) -> Montgomery-reduced UN limbs, value < 2p
    static ECGPU_HD E p_reduce(uint64_t* c) {
        if constexpr (C::ID == CURVE_P256) {
            // p256 in sparse form: u p = -u + u 2^96 + u 2^192 + u 2^224 (2^32 - 1).  The -u clears the low 28 bits
            // of c[i]; the other three terms are one multiply-add each into columns i+3, i+6, i+8 (3 per row
            // instead of 6 for the limb form of p; model and bounds: tools/field_model.py p256_reduce_rows).
#pragma unroll
            for (int i = 0; i < UN; i++) {
                uint32_t u = (uint32_t)c[i] & PMASK;
                c[i + 1] += (c[i] >> UB);
                c[i + 3] += (uint64_t)u * opaque_const(1u << 12);
                c[i + 6] += (uint64_t)u * opaque_const(1u << 24);
                c[i + 8] += (uint64_t)u * opaque_const(0xFFFFFFFFu);
            }
            E r;
            uint64_t v = c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (C::ID == CURVE_P384) {
            // p384 in sparse form with SIGNED columns: u p = -u + u 2^32 - u 2^96 - u 2^128 + u 2^384, i.e. + u 2^5
            // into column i+1 (with the arithmetic carry of c[i]), - u 2^15 into i+3, - u 2^20 into i+4, + u 2^6 into
            // i+14: 4 multiply-adds per row instead of 13 (tools/field_model.py p384_mont_mul; the sign bit is why
            // the product limit is 30).
#pragma unroll
            for (int i = 0; i < UN; i++) {
                const int64_t ci = (int64_t)c[i];
                const int64_t u = (int64_t)((uint32_t)ci & PMASK);
                c[i + 1] = (uint64_t)((int64_t)c[i + 1] + (ci >> UB) + u * (int64_t)(int32_t)opaque_const(1u << 5));
                c[i + 3] = (uint64_t)((int64_t)c[i + 3] + u * (int64_t)(int32_t)opaque_const(0u - (1u << 15)));
                c[i + 4] = (uint64_t)((int64_t)c[i + 4] + u * (int64_t)(int32_t)opaque_const(0u - (1u << 20)));
                c[i + 14] = (uint64_t)((int64_t)c[i + 14] + u * (int64_t)(int32_t)opaque_const(1u << 6));
            }
            E r;
            int64_t v = (int64_t)c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = (int64_t)c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (C::ID == CURVE_P521) {
            // p = 2^521 - 1: u p = u 2^521 - u.  The -u clears the low 27 bits of c[i]; u 2^521 = u 2^8 at limb i + 19.
#pragma unroll
            for (int i = 0; i < UN; i++) {
                uint32_t u = (uint32_t)c[i] & PMASK;
                c[i + 1] += (c[i] >> UB);
                c[i + 19] += (uint64_t)u << 8;
            }
            E r;
            uint64_t v = c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (C::ID == CURVE_SM2 || C::ID == CURVE_P224 || C::ID == CURVE_P192) {
            // Sparse rows on unsigned, BIASED columns (p_columns_init): u p as a few power-of-two terms, the negative ones as
            // signed multiply-adds that cannot take their column below zero because it started from the bias.
            //   sm2   p = 2^256 - 2^224 - 2^96 + 2^64 - 1:  -u (clears), + u 2^8 -> i+2, - u 2^12 -> i+3, + u (2^32 - 1) -> i+8
            //   p224  p = 2^224 - 2^96 + 1 (p = 1 mod 2^27, u = -c_i):  +u (clears), - u 2^15 -> i+3, + u 2^8 -> i+8
            //   p192  p = 2^192 - 2^64 - 1:  -u (clears), - u 2^12 -> i+2, + u 2^10 -> i+7
            // 3 / 2 / 2 multiply-adds per row instead of 9 / 6 / 7 for the limb form (model with overflow and sign checks:
            // tools/field_model.py sparse_mont_mul).
            static_assert(HasBias<PC>::value, "sparse rows need the column biases");
#pragma unroll
            for (int i = 0; i < UN; i++) {
                uint32_t u;
                if constexpr (C::ID == CURVE_P224) {
                    u = (0u - (uint32_t)c[i]) & PMASK;
                    c[i + 1] += (c[i] + u) >> UB;
                } else {
                    u = (uint32_t)c[i] & PMASK;
                    c[i + 1] += c[i] >> UB;
                }
                const int64_t su = (int64_t)u;
                if constexpr (C::ID == CURVE_SM2) {
                    c[i + 2] += (uint64_t)u * opaque_const(1u << 8);
                    c[i + 3] = (uint64_t)((int64_t)c[i + 3] + su * (int64_t)(int32_t)opaque_const(0u - (1u << 12)));
                    c[i + 8] += (uint64_t)u * opaque_const(0xFFFFFFFFu);
                } else if constexpr (C::ID == CURVE_P224) {
                    c[i + 3] = (uint64_t)((int64_t)c[i + 3] + su * (int64_t)(int32_t)opaque_const(0u - (1u << 15)));
                    c[i + 8] += (uint64_t)u * opaque_const(1u << 8);
                } else {
                    c[i + 2] = (uint64_t)((int64_t)c[i + 2] + su * (int64_t)(int32_t)opaque_const(0u - (1u << 12)));
                    c[i + 7] += (uint64_t)u * opaque_const(1u << 10);
                }
            }
            E r;
            uint64_t v = c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (C::ID == CURVE_BIGN256) {
            // bign-curve256v1: p = 2^256 - 189 in sparse form with SIGNED columns: u p = -189 u + u 2^256, i.e. -189 u into
            // column i (with u = c_i / 189 mod 2^28 it clears the low 28 bits) and + 16 u into column i + 9
            // (2^256 = 2^4 * 2^(9 * 28)): two multiply-adds per row instead of the ten of the limb form.  The sign bit is why
            // the product limit of this parameter set is 11 (tools/gen_field_consts.py).
            static_assert(UB == 28 && UN == 10, "bign256: 10 x 28 limbs");
#pragma unroll
            for (int i = 0; i < UN; i++) {
                const int64_t ci = (int64_t)c[i];
                const int64_t u = (int64_t)(((uint32_t)ci * PC::PINV) & PMASK);
                const int64_t t = ci + u * (int64_t)(int32_t)opaque_const(0u - 189u);
                c[i + 1] = (uint64_t)((int64_t)c[i + 1] + (t >> UB));
                c[i + 9] = (uint64_t)((int64_t)c[i + 9] + u * (int64_t)(int32_t)opaque_const(16u));
            }
            E r;
            int64_t v = (int64_t)c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = (int64_t)c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (!PC::P0_IS_MINUS_ONE && !PC::P0_IS_ONE) {
            // general p (brainpool): u = c_i * (-p^-1) mod 2^B, then c += u * p over all limbs; the low bits of c_i cancel
            // (model: tools/field_model.py umont_mul_general)
#pragma unroll
            for (int i = 0; i < UN; i++) {
                uint32_t u = ((uint32_t)c[i] * PC::PINV) & PMASK;
#pragma unroll
                for (int j = 0; j < UN; j++) {
                    if (PC::P[j] != 0) c[i + j] += (uint64_t)u * opaque_const(PC::P[j]);
                }
                c[i + 1] += c[i] >> UB;
            }
            E r;
            uint64_t v = c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
        if constexpr (PC::P0_IS_ONE) {
            // p = 1 (mod 2^B), as for p224: -p^-1 = -1, u = -c_i mod 2^B, and u * p0 = u clears the low bits of c_i
            // (model: tools/field_model.py umont_mul_general)
            static_assert(PC::P[0] == 1, "p0 = 1 expected");
#pragma unroll
            for (int i = 0; i < UN; i++) {
                uint32_t u = (0u - (uint32_t)c[i]) & PMASK;
                c[i + 1] += (c[i] + u) >> UB;
#pragma unroll
                for (int j = 1; j < UN; j++) {
                    if (PC::P[j] != 0) c[i + j] += (uint64_t)u * opaque_const(PC::P[j]);
                }
            }
            E r;
            uint64_t v = c[UN];
#pragma unroll
            for (int k = 0; k < UN - 1; k++) {
                r.v[k] = (uint32_t)v & PMASK;
                v = c[UN + 1 + k] + (v >> UB);
            }
            r.v[UN - 1] = (uint32_t)v;
            return r;
        }
#pragma unroll
        for (int i = 0; i < UN; i++) {
            uint32_t u = (uint32_t)c[i] & PMASK;
            // (c[i] + u * p0) >> B = (c[i] >> B) + u since p0 = 2^B - 1: merged into the p1 term
            c[i + 1] += (c[i] >> UB);
            c[i + 1] += (uint64_t)u * opaque_const(PC::P[1] + 1u);
#pragma unroll
            for (int j = 2; j < UN; j++) {
                if (PC::P[j] != 0) c[i + j] += (uint64_t)u * opaque_const(PC::P[j]);
            }
        }
        E r;
        uint64_t v = c[UN];
#pragma unroll
        for (int k = 0; k < UN - 1; k++) {
            r.v[k] = (uint32_t)v & PMASK;
            v = c[UN + 1 + k] + (v >> UB);
        }
        r.v[UN - 1] = (uint32_t)v;
        return r;
    }
    // the columns start from zero — or, for the parameter sets whose sparse reduction rows have negative terms (sm2, p224,
    // p192), from biases that keep every column non-negative and sum to a multiple of p (ecgpu_field_consts.h BIAS;
    // tools/field_model.py sparse_bias)
    template <class T, class = void>
    struct HasBias : std::false_type {};
    template <class T>
    struct HasBias<T, std::void_t<decltype(T::SPARSE_BIAS)>> : std::true_type {};
    static ECGPU_HD void p_columns_init(uint64_t* c) {
#pragma unroll
        for (int k = 0; k < 2 * UN + 1; k++) {
            if constexpr (HasBias<PC>::value) c[k] = PC::BIAS[k];
            else c[k] = 0;
        }
    }
    static ECGPU_HD void p_columns(uint64_t* c, const uint32_t* a, const uint32_t* b, bool accumulate) {
        if (!accumulate) p_columns_init(c);
#pragma unroll
        for (int i = 0; i < UN; i++) {
#pragma unroll
            for (int j = 0; j < UN; j++) c[i + j] += (uint64_t)a[i] * b[j];
        }
    }
    static ECGPU_HD void p_columns_sqr(uint64_t* c, const uint32_t* a) {
        uint32_t a2[UN];
        p_columns_init(c);
#pragma unroll
        for (int j = 0; j < UN; j++) a2[j] = a[j] << 1;
#pragma unroll
        for (int i = 0; i < UN; i++) {
            c[2 * i] += (uint64_t)a[i] * a[i];
#pragma unroll
            for (int j = i + 1; j < UN; j++) c[i + j] += (uint64_t)a[i] * a2[j];
        }
    }
    static ECGPU_HD E p_norm(const E& a) {             // carry propagation only; the value is unchanged
        E r;
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < UN - 1; k++) {
            uint32_t v = a.v[k] + carry;
            r.v[k] = v & PMASK;
            carry = v >> UB;
        }
        r.v[UN - 1] = a.v[UN - 1] + carry;
        return r;
    }
    // value in [0, 2p) with strict limbs -> [0, p)
    static ECGPU_HD E p_cond_sub(const E& a) {
        E d;
        int32_t borrow = 0;
#pragma unroll
        for (int k = 0; k < UN - 1; k++) {
            int32_t t = (int32_t)a.v[k] - (int32_t)PC::P[k] + borrow;
            d.v[k] = (uint32_t)t & PMASK;
            borrow = t >> UB;                          // 0 or -1
        }
        int32_t tt = (int32_t)a.v[UN - 1] - (int32_t)PC::P[UN - 1] + borrow;
        d.v[UN - 1] = (uint32_t)tt;
        bool lt = tt < 0;
        E r;
#pragma unroll
        for (int k = 0; k < UN; k++) r.v[k] = lt ? a.v[k] : d.v[k];
        return r;
    }
    static ECGPU_HD void p_limbs_to_words(uint32_t* w, const E& r) {   // strict limbs, value < 2^(32 N)
#pragma unroll
        for (int i = 0; i < N; i++) {
            int bit = 32 * i, l = bit / UB, sh = bit % UB;
            uint64_t x = (uint64_t)r.v[l] >> sh;
            if (l + 1 < UN) x |= (uint64_t)r.v[l + 1] << (UB - sh);
            if (l + 2 < UN) x |= (uint64_t)r.v[l + 2] << (2 * UB - sh);
            w[i] = (uint32_t)x;
        }
    }
    static ECGPU_HD E p_const(const uint32_t* limbs) {
        E r;
#pragma unroll
        for (int k = 0; k < UN; k++) r.v[k] = limbs[k];
        return r;
    }
    static ECGPU_HD E p_mont_mul(const E& a, const E& b) {
        uint64_t c[2 * UN + 1];
        p_columns(c, a.v, b.v, false);
        return p_reduce(c);
    }

    // =============================================================================================
    // the typed interface
    // =============================================================================================
    // Host-only debug build (-DECGPU_BOUNDS_CHECK, tests/hostcheck): verifies at run time that every element
    // really obeys the magnitudes its type declares — the run-time twin of the static_asserts, mirroring the
    // reference's debug-build magnitude checker (k256/src/arithmetic/field/field_impl.rs:17-22).
    template <int L, int V>
    static ECGPU_HD void check_mag(const E& e) {
#if defined(ECGPU_BOUNDS_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
        if constexpr (REPR == REPR_U29_K256) {
            for (int i = 0; i < NL; i++)
                if ((uint64_t)e.v[i] > (uint64_t)L * KC::LB) { __builtin_trap(); }
        } else if constexpr (REPR == REPR_U28_MONT) {
            for (int i = 0; i < NL - 1; i++)
                if ((uint64_t)e.v[i] > (uint64_t)L * PC::LB) { __builtin_trap(); }
            if ((uint64_t)e.v[NL - 1] > (uint64_t)V * PC::TOP1 + PC::TOP1 / 2) { __builtin_trap(); }
        }
#else
        (void)e;
#endif
    }
    template <int L, int V>
    static ECGPU_HD Mag<C, L, V> wrap(const E& e) {
        check_mag<L, V>(e);
        Mag<C, L, V> r;
        r.e = e;
        return r;
    }
    static ECGPU_HD M1 zero() {
        M1 r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.e.v[i] = 0;
        return r;
    }
    static ECGPU_HD M1 one() {
        M1 r = zero();
        if constexpr (REPR == REPR_U28_MONT) {
            r.e = p_const(PC::ONE);
        } else {
            r.e.v[0] = 1;
        }
        return r;
    }

    template <int LA, int VA, int LB, int VB>
    static ECGPU_HD auto add(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b) {
        static_assert(LA + LB <= MAXMAG, "limb magnitude overflow in add: normalise an operand first");
        Mag<C, LA + LB, VA + VB> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.e.v[i] = a.e.v[i] + b.e.v[i];
        check_mag<LA + LB, VA + VB>(r.e);
        return r;
    }
    template <int LA, int VA>
    static ECGPU_HD auto dbl(const Mag<C, LA, VA>& a) { return add(a, a); }

    // flag ? a : b, typed with the larger of the two magnitudes
    template <int LA, int VA, int LB, int VB>
    static ECGPU_HD auto sel(bool flag, const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b) {
        Mag<C, (LA > LB ? LA : LB), (VA > VB ? VA : VB)> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.e.v[i] = flag ? a.e.v[i] : b.e.v[i];
        return r;
    }

    // -b as (multiple of p) - b, limb-wise
    template <int LB, int VB>
    static ECGPU_HD auto neg(const Mag<C, LB, VB>& b) {
        if constexpr (REPR == REPR_U29_K256) {
            constexpr int m = LB > VB ? LB : VB;
            static_assert(m <= KC::ZMAX, "no subtraction constant for this magnitude: normalise first");
            Mag<C, m + 1, m + 1> r;
            check_mag<LB, VB>(b.e);
#pragma unroll
            for (int i = 0; i < NL; i++) r.e.v[i] = KC::Z[m][i] - b.e.v[i];
            check_mag<m + 1, m + 1>(r.e);
            return r;
        } else {
            static_assert(LB <= PC::LMAX && VB <= PC::VMAX, "no subtraction constant for this magnitude");
            Mag<C, LB + 1, VB + 1> r;
            check_mag<LB, VB>(b.e);
#pragma unroll
            for (int i = 0; i < NL; i++) r.e.v[i] = PC::Z[LB][VB][i] - b.e.v[i];
            check_mag<LB + 1, VB + 1>(r.e);
            return r;
        }
    }
    template <int LA, int VA, int LB, int VB>
    static ECGPU_HD auto sub(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b) {
        return add(a, neg(b));
    }

    template <int LA, int VA, int LB, int VB>
    static ECGPU_HD M1 mul(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b) {
        if constexpr (REPR == REPR_U29_K256) {
            static_assert(LA * LB <= MAXPROD, "k256 mul: limb magnitude product too large");
            uint64_t c[17];
            k_columns(c, a.e.v, b.e.v, false);
            return wrap<1, 1>(k_reduce(c));
        } else {
            static_assert(LA * LB <= MAXPROD, "p256 mul: limb magnitude product too large");
            static_assert((long)VA * VB <= (1L << PC::VLIMIT_LOG2), "mul: value magnitude product too large");
            uint64_t c[2 * UN + 1];
            p_columns(c, a.e.v, b.e.v, false);
            return wrap<1, 1>(p_reduce(c));
        }
    }
    template <int LA, int VA>
    static ECGPU_HD M1 sqr(const Mag<C, LA, VA>& a) {
        if constexpr (REPR == REPR_U29_K256) {
            static_assert(LA * LA <= MAXPROD, "k256 sqr: limb magnitude too large");
            uint64_t c[17];
            k_columns_sqr(c, a.e.v);
            return wrap<1, 1>(k_reduce(c));
        } else {
            static_assert(LA * LA <= MAXPROD, "p256 sqr: limb magnitude too large");
            uint64_t c[2 * UN + 1];
            p_columns_sqr(c, a.e.v);
            return wrap<1, 1>(p_reduce(c));
        }
    }
    // a*b + c*d with ONE reduction (the two products share their column accumulators)
    template <int LA, int VA, int LB, int VB, int LC, int VC, int LD, int VD>
    static ECGPU_HD M1 mul2(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b, const Mag<C, LC, VC>& c,
                            const Mag<C, LD, VD>& d) {
        if constexpr (REPR == REPR_U29_K256) {
            static_assert(LA * LB + LC * LD <= MAXPROD, "k256 mul2: limb magnitude products too large");
            uint64_t col[17];
            k_columns(col, a.e.v, b.e.v, false);
            k_columns(col, c.e.v, d.e.v, true);
            return wrap<1, 1>(k_reduce(col));
        } else {
            static_assert(LA * LB + LC * LD <= MAXPROD, "p256 mul2: limb magnitude products too large");
            static_assert((long)VA * VB + (long)VC * VD <= (1L << PC::VLIMIT_LOG2), "mul2: value magnitudes too large");
            uint64_t col[2 * UN + 1];
            p_columns(col, a.e.v, b.e.v, false);
            p_columns(col, c.e.v, d.e.v, true);
            return wrap<1, 1>(p_reduce(col));
        }
    }
    // a * b - c  and  a^2 - c  with ONE reduction and no carry pass of their own (round 4).
    // k256: the subtrahend enters the LOW product columns as (multiple of p) - c, limb by limb — one multiply-add by an opaque 1
    // each: a 64-bit addition would first need the 32-bit limb zero-extended into a register pair — and the carry pass of the
    // reduction normalises the difference for free, where norm(sub(mul(a, b), c)) pays a limb-wise subtraction plus a carry pass
    // of its own (9 issue slots less per use, three uses per mixed XYZZ addition).  The columns have the room: 9 * MAXPROD * LB^2
    // < 2^64 - 2^58, the subtrahend adds < 2^32 per column (model: tools/field_model.py k256_mul_sub).  Result magnitude (1, 1).
    // Montgomery fields: the reduction divides by R, so the subtrahend goes into the HIGH columns (UN + k: the ones the final
    // carry pass turns into the result) — the same multiply-add per limb, the same carry pass saved.  The result is what
    // norm(sub(mul(a, b), c)) returns: limb magnitude 1, value magnitude 1 + (VC + 1) (models: p256_mont_mul & co., `hi_add`).
    template <int LC, int VC>
    static ECGPU_HD void k_columns_sub(uint64_t* col, const Mag<C, LC, VC>& c) {
        constexpr int m = LC > VC ? LC : VC;
        static_assert(m <= KC::ZMAX, "no subtraction constant for this magnitude: normalise first");
        check_mag<LC, VC>(c.e);
        const uint32_t one = opaque_const(1u);
#pragma unroll
        for (int k = 0; k < 9; k++) col[k] += (uint64_t)(KC::Z[m][k] - c.e.v[k]) * one;
    }
    template <int LC, int VC>
    static ECGPU_HD void p_columns_sub(uint64_t* col, const Mag<C, LC, VC>& c) {
        static_assert(LC <= PC::LMAX && VC <= PC::VMAX, "no subtraction constant for this magnitude");
        check_mag<LC, VC>(c.e);
        const uint32_t one = opaque_const(1u);
#pragma unroll
        for (int k = 0; k < UN; k++) col[UN + k] += (uint64_t)(PC::Z[LC][VC][k] - c.e.v[k]) * one;
    }
    template <int LA, int VA, int LB, int VB, int LC, int VC>
    static ECGPU_HD auto mul_sub(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b, const Mag<C, LC, VC>& c) {
        if constexpr (!ECGPU_FUSED_SUB) {
            return norm(sub(mul(a, b), c));
        } else if constexpr (REPR == REPR_U29_K256) {
            static_assert(LA * LB <= MAXPROD, "k256 mul_sub: limb magnitude product too large");
            uint64_t col[17];
            k_columns(col, a.e.v, b.e.v, false);
            k_columns_sub(col, c);
            return wrap<1, 1>(k_reduce(col));
        } else {
            static_assert(LA * LB <= MAXPROD, "mul_sub: limb magnitude product too large");
            static_assert((long)VA * VB <= (1L << PC::VLIMIT_LOG2), "mul_sub: value magnitude product too large");
            uint64_t col[2 * UN + 1];
            p_columns(col, a.e.v, b.e.v, false);
            p_columns_sub(col, c);
            return wrap<1, VC + 2>(p_reduce(col));
        }
    }
    template <int LA, int VA, int LC, int VC>
    static ECGPU_HD auto sqr_sub(const Mag<C, LA, VA>& a, const Mag<C, LC, VC>& c) {
        if constexpr (!ECGPU_FUSED_SUB) {
            return norm(sub(sqr(a), c));
        } else if constexpr (REPR == REPR_U29_K256) {
            static_assert(LA * LA <= MAXPROD, "k256 sqr_sub: limb magnitude too large");
            uint64_t col[17];
            k_columns_sqr(col, a.e.v);
            k_columns_sub(col, c);
            return wrap<1, 1>(k_reduce(col));
        } else {
            static_assert(LA * LA <= MAXPROD, "sqr_sub: limb magnitude too large");
            static_assert((long)VA * VA <= (1L << PC::VLIMIT_LOG2), "sqr_sub: value magnitude too large");
            uint64_t col[2 * UN + 1];
            p_columns_sqr(col, a.e.v);
            p_columns_sub(col, c);
            return wrap<1, VC + 2>(p_reduce(col));
        }
    }
    // limb magnitude back to 1 (k256: value magnitude too; p256: the value is untouched)
    template <int LA, int VA>
    static ECGPU_HD auto norm(const Mag<C, LA, VA>& a) {
        if constexpr (REPR == REPR_U29_K256) return wrap<1, 1>(k_norm(a.e));
        else return wrap<1, VA>(p_norm(a.e));
    }
    // norm(a) only if its limb magnitude exceeds LIM (compile-time decision); SQLIM is the largest limb
    // magnitude that may still be squared
    ECGPU_CONST int SQLIM = MAXPROD >= 49 ? 7 : (MAXPROD >= 16 ? 4 : (MAXPROD >= 4 ? 2 : 1));
    template <int LIM, int LA, int VA>
    static ECGPU_HD auto fit(const Mag<C, LA, VA>& a) {
        if constexpr (LA > LIM) return norm(a);
        else return a;
    }
    // a * K for a small compile-time constant, result magnitude (1, 1); k256 only (b3 = 21 and friends)
    template <uint32_t K, int LA, int VA>
    static ECGPU_HD M1 mul_small(const Mag<C, LA, VA>& a) {
        static_assert(REPR == REPR_U29_K256, "mul_small is only provided for k256");
        static_assert(K < (1u << 12), "constant too large");
        return wrap<1, 1>(k_mul_small(a.e, K));
    }

    // ---- canonical words / bytes -----------------------------------------------------------------
    // canonical value (< p, checked by the caller) as N little-endian words -> internal form
    static ECGPU_HD M1 from_canonical(const uint32_t* w) {
        if constexpr (REPR == REPR_U29_K256) {
            return wrap<1, 1>(words_to_limbs<9, 29>(w));
        } else {
            E a = words_to_limbs<UN, UB>(w);
            return wrap<1, 1>(p_mont_mul(a, p_const(PC::R2)));
        }
    }
    template <int LA, int VA>
    static ECGPU_HD void to_canonical(uint32_t* w, const Mag<C, LA, VA>& a) {
        if constexpr (REPR == REPR_U29_K256) {
            check_mag<LA, VA>(a.e);
            k_to_words<(LA <= 1)>(w, a.e);
        } else {
            static_assert(LA <= MAXPROD, "normalise before to_canonical");
            E onep;                                       // plain 1: a * 1 * R^-1 leaves the Montgomery domain
#pragma unroll
            for (int k = 0; k < UN; k++) onep.v[k] = k == 0 ? 1u : 0u;
            E r = p_cond_sub(p_mont_mul(a.e, onep));
            p_limbs_to_words(w, r);
        }
    }
    // internal-domain value fully reduced to [0, p), N words ("packed" storage form: table entries, MSM points)
    template <int LA, int VA>
    static ECGPU_HD void pack(uint32_t* w, const Mag<C, LA, VA>& a) {
        if constexpr (REPR == REPR_U29_K256) {
            check_mag<LA, VA>(a.e);
            k_to_words<(LA <= 1)>(w, a.e);
        } else {
            static_assert(LA <= MAXPROD, "normalise before pack");
            E r = p_cond_sub(p_mont_mul(a.e, p_const(PC::ONE)));   // a * R * R^-1 = a, now < 2p with strict limbs
            p_limbs_to_words(w, r);
        }
    }
    static ECGPU_HD M1 unpack(const uint32_t* w) {
        if constexpr (REPR == REPR_U29_K256) {
            return wrap<1, 1>(words_to_limbs<9, 29>(w));
        } else {
            return wrap<1, 1>(words_to_limbs<UN, UB>(w));
        }
    }

    template <int LA, int VA>
    static ECGPU_HD bool is_zero(const Mag<C, LA, VA>& a) {
        uint32_t w[N];
        pack(w, a);
        return mp_is_zero<N>(w);
    }
    template <int LA, int VA, int LB, int VB>
    static ECGPU_HD bool eq(const Mag<C, LA, VA>& a, const Mag<C, LB, VB>& b) {
        return is_zero(norm(sub(a, b)));
    }

    // big-endian canonical bytes <-> internal; `ok` false if the encoded value is >= p
    static ECGPU_HD M1 from_bytes(const uint8_t* be, bool* ok) {
        uint32_t w[N];
        load_be_wire<C>(w, be);
        *ok = !mp_geq<N>(w, C::P);
        return from_canonical(w);
    }
    template <int LA, int VA>
    static ECGPU_HD void to_bytes(uint8_t* be, const Mag<C, LA, VA>& a) {
        uint32_t w[N];
        to_canonical(w, a);
        store_be_wire<C>(be, w);
    }

    static ECGPU_HD M1 sqr_n(M1 x, int n) {
#pragma unroll 1
        for (int i = 0; i < n; i++) x = sqr(x);
        return x;
    }
    // 1/a; a == 0 -> 0.  Like the reference (crypto-bigint's safegcd — k256 field.rs:178-184, primefield
    // monty.rs:373-375) this runs Bernstein–Yang division steps (ecgpu_modinv.h) on the canonical value; for the
    // Montgomery fields (aR)^-1 is brought back to a^-1 R by two multiplications with R^2.
    static ECGPU_HD M1 inv(const M1& a) {
        uint32_t w[N], r[N];
        pack(w, a);
        ModInv<N>::invert(r, w, C::P);
        M1 x = unpack(r);
        if constexpr (REPR == REPR_U28_MONT) {
            M1 r2 = wrap<1, 1>(p_const(PC::R2));
            x = mul(mul(x, r2), r2);
        }
        return x;
    }
    // a^(p-2), the first implementation, kept as an independent check of `inv` (tests/hostcheck).  k256 and p256
    // use addition chains over the runs of ones of p-2 (255 squarings + 15 resp. 12 multiplications); p384 a fixed
    // 4-bit window over the constant exponent (384 squarings + <= 110 multiplications).
    static ECGPU_HD M1 inv_fermat(const M1& a) {
        if constexpr (REPR == REPR_U29_K256) {
            // p-2 = 2^256 - 2^32 - 979: 223 ones, 0, 22 ones, 0000 1 0 11 0 1
            M1 x2 = mul(sqr(a), a);
            M1 x3 = mul(sqr(x2), a);
            M1 x6 = mul(sqr_n(x3, 3), x3);
            M1 x9 = mul(sqr_n(x6, 3), x3);
            M1 x11 = mul(sqr_n(x9, 2), x2);
            M1 x22 = mul(sqr_n(x11, 11), x11);
            M1 x44 = mul(sqr_n(x22, 22), x22);
            M1 x88 = mul(sqr_n(x44, 44), x44);
            M1 x176 = mul(sqr_n(x88, 88), x88);
            M1 x220 = mul(sqr_n(x176, 44), x44);
            M1 x223 = mul(sqr_n(x220, 3), x3);
            M1 t = mul(sqr_n(x223, 23), x22);
            t = mul(sqr_n(t, 5), a);
            t = mul(sqr_n(t, 3), x2);
            t = mul(sqr_n(t, 2), a);
            return t;
        } else if constexpr (REPR == REPR_U28_MONT && C::ID == CURVE_P256) {
            // p-2 = ffffffff 00000001 00000000 00000000 00000000 ffffffff ffffffff fffffffd
            M1 x2 = mul(sqr(a), a);
            M1 x3 = mul(sqr(x2), a);
            M1 x6 = mul(sqr_n(x3, 3), x3);
            M1 x12 = mul(sqr_n(x6, 6), x6);
            M1 x15 = mul(sqr_n(x12, 3), x3);
            M1 x30 = mul(sqr_n(x15, 15), x15);
            M1 x32 = mul(sqr_n(x30, 2), x2);
            M1 t = mul(sqr_n(x32, 32), a);
            t = mul(sqr_n(t, 128), x32);
            t = mul(sqr_n(t, 32), x32);
            t = mul(sqr_n(t, 30), x30);
            t = mul(sqr_n(t, 2), a);
            return t;
        } else {
            M1 tab[16];
            tab[0] = one();
            tab[1] = a;
#pragma unroll 1
            for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
            uint32_t e[N];
#pragma unroll
            for (int i = 0; i < N; i++) e[i] = C::P[i];
            {            // e = p - 2 (the low word of p224's p is 1: the borrow runs up)
                uint32_t borrow = 2;
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const uint32_t v = e[i];
                    e[i] = v - borrow;
                    borrow = v < borrow ? 1u : 0u;
                }
            }
            M1 r = one();
#pragma unroll 1
            for (int i = 8 * N - 1; i >= 0; i--) {
                uint32_t nib = (e[i >> 3] >> ((i & 7) * 4)) & 0xF;
                r = sqr(sqr(sqr(sqr(r))));
                if (nib) r = mul(r, tab[nib]);
            }
            return r;
        }
    }
    // p = 1 (mod 4) (p224: p - 1 = 2^96 (2^128 - 1)): Tonelli-Shanks in its fixed-schedule form.  With q = 2^128 - 1 and
    // g = z^q a generator of the subgroup of order 2^96 (z = 11, the smallest non-residue), x = a^((q+1)/2) has
    // x^2 = a t with t = a^q = g^e in that subgroup, e even exactly when a is a square; the bits of e come out one per
    // round (the lowest unknown bit is set iff t^(2^(95-i)) = -1; t is then multiplied by g^(-2^i) to clear it), and the
    // root is x g^(-e/2).  Every lane runs the same 96 rounds (4,560 squarings in all — against 223 for the 3 mod 4 primes —
    // and at most 190 multiplications): no lane waits for another's iteration count.  The reference delegates to
    // crypto-bigint (primefield/src/monty.rs:467-469, un-vendored); either root serves, the callers pick by parity.
    static ECGPU_HD M1 sqrt_sylow(const M1& a, bool* ok) {
        static_assert(C::ID == CURVE_P224, "Tonelli-Shanks constants exist for p224 only");
        constexpr int S = C::UC::SYLOW_S;
        // a^(2^127 - 1): 127 ones
        M1 x2 = mul(sqr(a), a);
        M1 x3 = mul(sqr(x2), a);
        M1 x6 = mul(sqr_n(x3, 3), x3);
        M1 x12 = mul(sqr_n(x6, 6), x6);
        M1 x24 = mul(sqr_n(x12, 12), x12);
        M1 x48 = mul(sqr_n(x24, 24), x24);
        M1 x96 = mul(sqr_n(x48, 48), x48);
        M1 x120 = mul(sqr_n(x96, 24), x24);
        M1 x126 = mul(sqr_n(x120, 6), x6);
        M1 w = mul(sqr(x126), a);
        M1 r = mul(w, a);                       // a^(2^127) = a^((q+1)/2)
        M1 t = mul(r, w);                       // a^(2^128 - 1) = a^q
        M1 gpow = wrap<1, 1>(p_const(C::UC::SYLOW_GINV));       // g^(-2^i)
        M1 ghalf = one();                       // g^(-2^(i-1)) (unused in round 0: an odd e means "no root")
#pragma unroll 1
        for (int i = 0; i < S; i++) {
            const M1 u = sqr_n(t, S - 1 - i);
            const bool bit = !eq(u, one());
            const M1 tg = mul(t, gpow), rg = mul(r, ghalf);
            t = wrap<1, 1>(sel(bit, tg, t).e);
            r = wrap<1, 1>(sel(bit && i > 0, rg, r).e);
            ghalf = gpow;
            gpow = sqr(gpow);
        }
        *ok = eq(sqr(r), a);
        return r;
    }
    // square root: a^((p+1)/4) (the primes that are 3 mod 4), *ok = whether the result squares back to a.
    // k256/src/arithmetic/field.rs:200-235 and p256/src/arithmetic/field.rs:121-147 (the same addition chains);
    // p384 delegates to crypto-bigint (primefield/src/monty.rs:467-469): a fixed 4-bit window over the exponent.
    static ECGPU_HD M1 sqrt(const M1& a, bool* ok) {
        M1 r;
        if constexpr (C::ID == CURVE_P224) {
            return sqrt_sylow(a, ok);
        } else if constexpr (REPR == REPR_U29_K256) {
            M1 x2 = mul(sqr(a), a);
            M1 x3 = mul(sqr(x2), a);
            M1 x6 = mul(sqr_n(x3, 3), x3);
            M1 x9 = mul(sqr_n(x6, 3), x3);
            M1 x11 = mul(sqr_n(x9, 2), x2);
            M1 x22 = mul(sqr_n(x11, 11), x11);
            M1 x44 = mul(sqr_n(x22, 22), x22);
            M1 x88 = mul(sqr_n(x44, 44), x44);
            M1 x176 = mul(sqr_n(x88, 88), x88);
            M1 x220 = mul(sqr_n(x176, 44), x44);
            M1 x223 = mul(sqr_n(x220, 3), x3);
            r = mul(sqr_n(x223, 23), x22);
            r = mul(sqr_n(r, 6), x2);
            r = sqr_n(r, 2);
        } else if constexpr (REPR == REPR_U28_MONT && C::ID == CURVE_P256) {
            M1 t11 = mul(a, sqr(a));
            M1 t1111 = mul(t11, sqr_n(t11, 2));
            M1 t8 = mul(t1111, sqr_n(t1111, 4));
            M1 x16 = mul(sqr_n(t8, 8), t8);
            r = mul(sqr_n(x16, 16), x16);
            r = mul(sqr_n(r, 32), a);
            r = mul(sqr_n(r, 96), a);
            r = sqr_n(r, 94);
        } else if constexpr (C::ID == CURVE_P521) {
            // (p + 1) / 4 = 2^519: squarings only.  (Also keeps the window table — a dynamically indexed private array,
            // i.e. scratch memory — out of the 20-limb kernels: k_ecdsa_recover_prepare<P521Params> with the table AND
            // AGPR-parked registers computed wrong scalars on gfx950 while the host build of the same source was right;
            // profiles/r02/diag_recover_p521.txt.)
            r = sqr_n(a, 519);
        } else {
            M1 tab[16];
            tab[0] = one();
            tab[1] = a;
#pragma unroll 1
            for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
            uint32_t e[N];                               // (p + 1) / 4; p + 1 does not carry out for these primes
            {
                uint64_t c = 1;
#pragma unroll
                for (int i = 0; i < N; i++) {
                    c += C::P[i];
                    e[i] = (uint32_t)c;
                    c >>= 32;
                }
#pragma unroll
                for (int i = 0; i < N; i++) e[i] = (e[i] >> 2) | (i + 1 < N ? e[i + 1] << 30 : 0u);
            }
            r = one();
#pragma unroll 1
            for (int i = 8 * N - 1; i >= 0; i--) {
                uint32_t nib = (e[i >> 3] >> ((i & 7) * 4)) & 0xF;
                r = sqr(sqr(sqr(sqr(r))));
                if (nib) r = mul(r, tab[nib]);
            }
        }
        *ok = eq(sqr(r), a);
        return r;
    }
    // parity of the canonical value
    template <int LA, int VA>
    static ECGPU_HD bool is_odd(const Mag<C, LA, VA>& a) {
        uint32_t w[N];
        to_canonical(w, a);
        return (w[0] & 1u) != 0;
    }
};

}  // namespace ecgpu
