// ecgpu_field.h — prime-field arithmetic for the MI355X scalar-mul engine.
//
// Every function is `__host__ __device__` so that the very same code the gfx950 kernels run can
// be compiled with g++ and unit-checked on a CPU against the oracle (tests/hostcheck/); the
// product library only ever instantiates it inside HIP kernels.
//
// Representation (one thread = one field element held in VGPRs, 32-bit limbs, little-endian):
//   k256  8 limbs, plain residues kept "weakly reduced" in [0, 2^256) and folded with
//         2^256 = 0x1000003D1 (mod p).  The reference's 64-bit build uses 5x52 lazy limbs
//         (k256/src/arithmetic/field/field_5x52.rs:240-401) and its 32-bit build 10x26
//         (field_10x26.rs:308-620); only canonical bytes are compared, so the GPU is free to
//         pick the layout that suits v_mad_u64_u32.
//   p256  8 limbs, Montgomery form R = 2^256, fully reduced — p256/src/arithmetic/field.rs:99-108,
//         field/field64.rs:83-123 (p' = 1 word-by-word reduction; same trick holds for 32-bit
//         words, field/field32.rs:110-208).
//   p384  12 limbs, Montgomery form R = 2^384, fully reduced — p384/src/arithmetic/field.rs:52-57
//         -> primefield/src/monty.rs:316-368 -> crypto-bigint ConstMontyForm (p' = 1 for 32-bit
//         words as well because p = -1 mod 2^32).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ECGPU_HD __host__ __device__ __forceinline__
#define ECGPU_CONST static constexpr
#else
#define ECGPU_HD inline
#define ECGPU_CONST static constexpr
#endif

namespace ecgpu {

enum CurveId : int { CURVE_K256 = 0, CURVE_P256 = 1, CURVE_P384 = 2 };

template <int N>
struct Fe {
    uint32_t v[N];
};

// ---------------------------------------------------------------------------------------------
// small multi-limb helpers
// ---------------------------------------------------------------------------------------------

template <int N>
ECGPU_HD uint32_t mp_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)a[i] + b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}

template <int N>
ECGPU_HD uint32_t mp_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (int64_t)a[i] - (int64_t)b[i];
        r[i] = (uint32_t)c;
        c >>= 32;  // arithmetic shift: 0 or -1
    }
    return (uint32_t)(c & 1);
}

// returns 1 if a >= b
template <int N>
ECGPU_HD bool mp_geq(const uint32_t* a, const uint32_t* b) {
    uint32_t t[N];
    return mp_sub<N>(t, a, b) == 0;
}

template <int N>
ECGPU_HD bool mp_is_zero(const uint32_t* a) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; i++) z |= a[i];
    return z == 0;
}

// r[0..2N) = a * b, operand scanning; one v_mad_u64_u32 per limb pair
template <int N>
ECGPU_HD void mp_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
#pragma unroll
    for (int i = 0; i < 2 * N; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + carry;
            r[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        r[i + N] = carry;
    }
}

// r[0..2N) = a^2: off-diagonal products once, doubled, plus the diagonal
template <int N>
ECGPU_HD void mp_sqr(uint32_t* r, const uint32_t* a) {
#pragma unroll
    for (int i = 0; i < 2 * N; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            uint64_t t = (uint64_t)a[i] * a[j] + r[i + j] + carry;
            r[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        r[i + N] = carry;
    }
    // double
    uint32_t top = 0;
#pragma unroll
    for (int i = 1; i < 2 * N; i++) {
        uint32_t w = r[i];
        r[i] = (w << 1) | top;
        top = w >> 31;
    }
    // add squares on the diagonal
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t sq = (uint64_t)a[i] * a[i];
        c += (uint64_t)r[2 * i] + (uint32_t)sq;
        r[2 * i] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)r[2 * i + 1] + (uint32_t)(sq >> 32);
        r[2 * i + 1] = (uint32_t)c;
        c >>= 32;
    }
}

ECGPU_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// big-endian bytes (4-byte aligned) -> little-endian limbs
template <int N>
ECGPU_HD void load_be(uint32_t* limbs, const uint8_t* bytes) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes);
#pragma unroll
    for (int i = 0; i < N; i++) limbs[i] = bswap32(w[N - 1 - i]);
}
template <int N>
ECGPU_HD void store_be(uint8_t* bytes, const uint32_t* limbs) {
    uint32_t* w = reinterpret_cast<uint32_t*>(bytes);
#pragma unroll
    for (int i = 0; i < N; i++) w[N - 1 - i] = bswap32(limbs[i]);
}

// ---------------------------------------------------------------------------------------------
// curve parameter packs (constants: SURVEY.md Appendix A, reference lines cited there)
// ---------------------------------------------------------------------------------------------

struct K256Params {
    ECGPU_CONST int ID = CURVE_K256;
    ECGPU_CONST int N = 8;            // 32-bit limbs per field element / scalar
    ECGPU_CONST bool A_IS_ZERO = true;
    ECGPU_CONST bool MONTGOMERY = false;
    // p = 2^256 - 0x1000003D1                      k256/src/arithmetic/field.rs:41-42
    ECGPU_CONST uint32_t P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order n                                k256/src/lib.rs:71
    ECGPU_CONST uint32_t ORDER[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                     0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // generator, canonical little-endian limbs     k256/src/arithmetic/affine.rs:65-79
    ECGPU_CONST uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                                  0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    ECGPU_CONST uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                                  0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    ECGPU_CONST uint32_t B_SMALL = 7;  // y^2 = x^3 + 7   k256/src/arithmetic.rs
};

struct P256Params {
    ECGPU_CONST int ID = CURVE_P256;
    ECGPU_CONST int N = 8;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p256/src/arithmetic.rs:44
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^256 - 2^224 + 2^192 + 2^96 - 1         p256/src/arithmetic/field.rs:35
    ECGPU_CONST uint32_t P[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u,
                                 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu};
    // n                                            p256/src/lib.rs:60
    ECGPU_CONST uint32_t ORDER[8] = {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu,
                                     0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu};
    // R^2 mod p, R = 2^256                         p256/src/arithmetic/field.rs:183-185
    ECGPU_CONST uint32_t R2[8] = {0x00000003u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFBu,
                                  0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFDu, 0x00000004u};
    // R mod p = 2^256 - p
    ECGPU_CONST uint32_t ONE[8] = {0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu,
                                   0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x00000000u};
    // curve b, canonical                           p256/src/arithmetic.rs:55-57
    ECGPU_CONST uint32_t B[8] = {0x27D2604Bu, 0x3BCE3C3Eu, 0xCC53B0F6u, 0x651D06B0u,
                                 0x769886BCu, 0xB3EBBD55u, 0xAA3A93E7u, 0x5AC635D8u};
    // generator, canonical                         p256/src/arithmetic.rs:67-74
    ECGPU_CONST uint32_t GX[8] = {0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u,
                                  0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u};
    ECGPU_CONST uint32_t GY[8] = {0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u,
                                  0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u};
};

struct P384Params {
    ECGPU_CONST int ID = CURVE_P384;
    ECGPU_CONST int N = 12;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p384/src/arithmetic.rs:44
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^384 - 2^128 - 2^96 + 2^32 - 1          p384/src/arithmetic/field.rs:34
    ECGPU_CONST uint32_t P[12] = {0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0xFFFFFFFFu,
                                  0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                  0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // n                                            p384/src/lib.rs:14
    ECGPU_CONST uint32_t ORDER[12] = {0xCCC52973u, 0xECEC196Au, 0x48B0A77Au, 0x581A0DB2u,
                                      0xF4372DDFu, 0xC7634D81u, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                      0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // R^2 mod p, R = 2^384: (2^128 + 2^96 - 2^32 + 1)^2
    ECGPU_CONST uint32_t R2[12] = {0x00000001u, 0xFFFFFFFEu, 0x00000000u, 0x00000002u,
                                   0x00000000u, 0xFFFFFFFEu, 0x00000000u, 0x00000002u,
                                   0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u};
    // R mod p = 2^128 + 2^96 - 2^32 + 1
    ECGPU_CONST uint32_t ONE[12] = {0x00000001u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u,
                                    0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u,
                                    0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u};
    // curve b, canonical                           p384/src/arithmetic.rs:57-59
    ECGPU_CONST uint32_t B[12] = {0xD3EC2AEFu, 0x2A85C8EDu, 0x8A2ED19Du, 0xC656398Du,
                                  0x5013875Au, 0x0314088Fu, 0xFE814112u, 0x181D9C6Eu,
                                  0xE3F82D19u, 0x988E056Bu, 0xE23EE7E4u, 0xB3312FA7u};
    // generator, canonical                         p384/src/arithmetic.rs:71-78
    ECGPU_CONST uint32_t GX[12] = {0x72760AB7u, 0x3A545E38u, 0xBF55296Cu, 0x5502F25Du,
                                   0x82542A38u, 0x59F741E0u, 0x8BA79B98u, 0x6E1D3B62u,
                                   0xF320AD74u, 0x8EB1C71Eu, 0xBE8B0537u, 0xAA87CA22u};
    ECGPU_CONST uint32_t GY[12] = {0x90EA0E5Fu, 0x7A431D7Cu, 0x1D7E819Du, 0x0A60B1CEu,
                                   0xB5F0B8C0u, 0xE9DA3113u, 0x289A147Cu, 0xF8F41DBDu,
                                   0x9292DC29u, 0x5D9E98BFu, 0x96262C6Fu, 0x3617DE4Au};
};

// ---------------------------------------------------------------------------------------------
// Field<C>
// ---------------------------------------------------------------------------------------------

template <class C>
struct Field {
    ECGPU_CONST int N = C::N;
    using E = Fe<C::N>;

    static ECGPU_HD E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    static ECGPU_HD E one() {
        E r = zero();
        if constexpr (C::MONTGOMERY) {
#pragma unroll
            for (int i = 0; i < N; i++) r.v[i] = C::ONE[i];
        } else {
            r.v[0] = 1;
        }
        return r;
    }

    // ---- k256: fold bits >= 2^256 with 2^256 = 2^32 + 977 (mod p) ----------------------------
    // value = lo[0..8) + top * 2^256, top < 2^34  ->  weakly reduced 8 limbs
    static ECGPU_HD void k256_fold_top(uint32_t* r, uint64_t top) {
        uint64_t c = (uint64_t)r[0] + (top & 0xFFFFFFFFu) * 977u;
        r[0] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)r[1] + (top & 0xFFFFFFFFu) + (top >> 32) * 977u;
        r[1] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)r[2] + (top >> 32);
        r[2] = (uint32_t)c;
        c >>= 32;
#pragma unroll
        for (int i = 3; i < 8; i++) {
            c += r[i];
            r[i] = (uint32_t)c;
            c >>= 32;
        }
        if (c) {  // wrapped past 2^256 once more: the low part is tiny, one more fold cannot wrap
            uint64_t d = (uint64_t)r[0] + 977u;
            r[0] = (uint32_t)d;
            d >>= 32;
            d += (uint64_t)r[1] + 1u;
            r[1] = (uint32_t)d;
            d >>= 32;
#pragma unroll
            for (int i = 2; i < 8; i++) {
                d += r[i];
                r[i] = (uint32_t)d;
                d >>= 32;
            }
        }
    }

    // 512-bit t -> weakly reduced 256-bit
    static ECGPU_HD E k256_reduce_wide(const uint32_t* t) {
        E r;
        // r = lo + hi*977 + (hi << 32)
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)t[8 + i] * 977u + t[i];
            if (i > 0) c += t[8 + i - 1];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
        uint64_t top = c + t[15];
        k256_fold_top(r.v, top);
        return r;
    }

    // ---- Montgomery reduction for p' = 1 (u = t[i]) ------------------------------------------
    static ECGPU_HD E mont_reduce_wide(uint32_t* t) {
        uint32_t top = 0;  // carry waiting to enter limb i+N of the next round
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t u = t[i];
            uint32_t carry = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                uint64_t s = (uint64_t)u * C::P[j] + t[i + j] + carry;
                t[i + j] = (uint32_t)s;
                carry = (uint32_t)(s >> 32);
            }
            uint64_t s = (uint64_t)t[i + N] + carry + top;
            t[i + N] = (uint32_t)s;
            top = (uint32_t)(s >> 32);
        }
        E r;
        uint32_t d[N];
        uint32_t borrow = mp_sub<N>(d, t + N, C::P);
        bool use_d = top || !borrow;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = use_d ? d[i] : t[N + i];
        return r;
    }

    static ECGPU_HD E mul(const E& a, const E& b) {
        uint32_t t[2 * N];
        mp_mul<N>(t, a.v, b.v);
        if constexpr (C::MONTGOMERY) return mont_reduce_wide(t);
        else return k256_reduce_wide(t);
    }
    static ECGPU_HD E sqr(const E& a) {
        uint32_t t[2 * N];
        mp_sqr<N>(t, a.v);
        if constexpr (C::MONTGOMERY) return mont_reduce_wide(t);
        else return k256_reduce_wide(t);
    }

    static ECGPU_HD E add(const E& a, const E& b) {
        E r;
        uint32_t c = mp_add<N>(r.v, a.v, b.v);
        if constexpr (C::MONTGOMERY) {
            uint32_t d[N];
            uint32_t borrow = mp_sub<N>(d, r.v, C::P);
            bool use_d = c || !borrow;
#pragma unroll
            for (int i = 0; i < N; i++) r.v[i] = use_d ? d[i] : r.v[i];
        } else {
            if (c) k256_fold_top(r.v, 1);
        }
        return r;
    }
    static ECGPU_HD E sub(const E& a, const E& b) {
        E r;
        uint32_t borrow = mp_sub<N>(r.v, a.v, b.v);
        if constexpr (C::MONTGOMERY) {
            if (borrow) mp_add<N>(r.v, r.v, C::P);
        } else {
            // wrapped result is a - b + 2^256 = a - b + 0x1000003D1 (mod p): take the excess off
            if (borrow) {
                int64_t c = (int64_t)r.v[0] - 977;
                r.v[0] = (uint32_t)c;
                c >>= 32;
                c += (int64_t)r.v[1] - 1;
                r.v[1] = (uint32_t)c;
                c >>= 32;
#pragma unroll
                for (int i = 2; i < 8; i++) {
                    c += r.v[i];
                    r.v[i] = (uint32_t)c;
                    c >>= 32;
                }
                if (c) {  // went below zero again (a - b + 2^256 < 0x1000003D1): add p back
                    mp_add<N>(r.v, r.v, C::P);
                }
            }
        }
        return r;
    }
    static ECGPU_HD E neg(const E& a) { return sub(zero(), a); }
    static ECGPU_HD E dbl(const E& a) { return add(a, a); }

    // a * k for a small constant k (k < 2^16)
    static ECGPU_HD E mul_small(const E& a, uint32_t k) {
        if constexpr (C::MONTGOMERY) {
            // not on the hot path for a = -3 curves; plain double-and-add
            E r = zero(), base = a;
            for (; k; k >>= 1) {
                if (k & 1) r = add(r, base);
                base = dbl(base);
            }
            return r;
        } else {
            E r;
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                c += (uint64_t)a.v[i] * k;
                r.v[i] = (uint32_t)c;
                c >>= 32;
            }
            k256_fold_top(r.v, c);
            return r;
        }
    }

    // canonical (fully reduced, non-Montgomery) limbs
    static ECGPU_HD E to_canonical(const E& a) {
        if constexpr (C::MONTGOMERY) {
            uint32_t t[2 * N];
#pragma unroll
            for (int i = 0; i < N; i++) { t[i] = a.v[i]; t[N + i] = 0; }
            return mont_reduce_wide(t);
        } else {
            E r = a;
            uint32_t d[N];
            if (mp_sub<N>(d, a.v, C::P) == 0) {
#pragma unroll
                for (int i = 0; i < N; i++) r.v[i] = d[i];
            }
            return r;
        }
    }
    // canonical limbs (must be < p) -> internal form
    static ECGPU_HD E from_canonical(const E& a) {
        if constexpr (C::MONTGOMERY) {
            E r2;
#pragma unroll
            for (int i = 0; i < N; i++) r2.v[i] = C::R2[i];
            return mul(a, r2);
        } else {
            return a;
        }
    }
    static ECGPU_HD bool is_zero(const E& a) {
        if constexpr (C::MONTGOMERY) {
            return mp_is_zero<N>(a.v);
        } else {
            uint32_t x = 0, y = 0;
#pragma unroll
            for (int i = 0; i < N; i++) { x |= a.v[i]; y |= a.v[i] ^ C::P[i]; }
            return x == 0 || y == 0;
        }
    }
    static ECGPU_HD bool eq(const E& a, const E& b) { return is_zero(sub(a, b)); }

    // big-endian canonical bytes <-> internal; `ok` false if the encoded value is >= p
    static ECGPU_HD E from_bytes(const uint8_t* be, bool* ok) {
        E c;
        load_be<N>(c.v, be);
        *ok = !mp_geq<N>(c.v, C::P);
        return from_canonical(c);
    }
    static ECGPU_HD void to_bytes(uint8_t* be, const E& a) {
        E c = to_canonical(a);
        store_be<N>(be, c.v);
    }

    // a^(p-2) by a fixed 4-bit window over the constant exponent; a == 0 -> 0.
    // (The reference inverts with crypto-bigint's safegcd — k256 field.rs:178-184,
    // primefield monty.rs:373-375; the inverse is unique so any method agrees.)
    static ECGPU_HD E inv(const E& a) {
        E tab[16];
        tab[0] = one();
        tab[1] = a;
#pragma unroll 1
        for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = C::P[i];
        e[0] -= 2;  // p is odd and p[0] >= 2 for all three curves
        E r = one();
#pragma unroll 1
        for (int i = 8 * N - 1; i >= 0; i--) {
            uint32_t nib = (e[i >> 3] >> ((i & 7) * 4)) & 0xF;
            r = sqr(r); r = sqr(r); r = sqr(r); r = sqr(r);
            if (nib) r = mul(r, tab[nib]);
        }
        return r;
    }
};

}  // namespace ecgpu
