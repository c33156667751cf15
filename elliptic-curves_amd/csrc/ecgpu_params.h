// ecgpu_params.h — limb helpers and curve parameter packs shared by host and device code.
// (Split out of ecgpu_field.h; constants: SURVEY.md Appendix A, reference lines cited per constant.)
#pragma once
#include <type_traits>

#include <stdint.h>

#include "ecgpu_field_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ECGPU_HD __host__ __device__ __forceinline__
#define ECGPU_CONST static constexpr
#else
#define ECGPU_HD inline
#define ECGPU_CONST static constexpr
#endif

namespace ecgpu {

// window width of the uniform-schedule generator LUTs (ecgpu_ctmul.h `fixed_base_mul_ct`; built by ensure_ct_lut in ecgpu_api.hip)
constexpr int CT_BASE_W = 6;
constexpr int CT_BASE_ENTRIES = 1 << (CT_BASE_W - 1);


enum CurveId : int { CURVE_K256 = 0, CURVE_P256 = 1, CURVE_P384 = 2, CURVE_SM2 = 3, CURVE_P224 = 4, CURVE_P192 = 5, CURVE_P521 = 6, CURVE_BP256 = 7, CURVE_BP384 = 8, CURVE_BP256T1 = 9, CURVE_BP384T1 = 10, CURVE_BIGN256 = 11 };

// in-register field representations (ecgpu_field.h)
enum Repr : int {
    REPR_U29_K256 = 1,   // 9 x 29-bit limbs, plain residues, lazily reduced, 2^261 folding (k256)
    REPR_U28_MONT = 2    // unsaturated limbs, Montgomery form, lazily reduced: 10 x 28 bit (p256), 15 x 27 bit (p384)
};

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
    ECGPU_CONST int N = 8;            // 32-bit words per canonical field element / scalar
    ECGPU_CONST int NL = 9;           // limbs held in registers
    ECGPU_CONST int REPR = REPR_U29_K256;
    using UC = consts::P256U;         // (unused for k256; keeps the field template well-formed)
    ECGPU_CONST bool A_IS_ZERO = true;
    ECGPU_CONST bool MONTGOMERY = false;
    // p = 2^256 - 0x1000003D1                      k256/src/arithmetic/field.rs:41-42
    ECGPU_CONST uint32_t P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order n                                k256/src/lib.rs:71
    ECGPU_CONST uint32_t ORDER[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                     0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order in Montgomery form (R = 2^256): R^2 mod n and -n^-1 mod 2^32, for the scalar arithmetic of the
    // ECDSA verification path (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[8] = {0x67D7D140u, 0x896CF214u, 0x0E7CF878u, 0x741496C2u,
                                           0x5BCD07C6u, 0xE697F5E4u, 0x81C69BC5u, 0x9D671CD5u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x5588B13Fu;
    // generator, canonical little-endian limbs     k256/src/arithmetic/affine.rs:65-79
    ECGPU_CONST uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu,
                                  0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    ECGPU_CONST uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u,
                                  0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    ECGPU_CONST uint32_t B_SMALL = 7;  // y^2 = x^3 + 7   k256/src/arithmetic.rs
    // beta: lambda * (x, y) = (beta x, y)            k256/src/arithmetic/projective.rs:31-37
    ECGPU_CONST uint32_t BETA[8] = {0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u,
                                    0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu};
};

struct P256Params {
    ECGPU_CONST int ID = CURVE_P256;
    ECGPU_CONST int N = 8;
    ECGPU_CONST int NL = 10;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::P256U;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p256/src/arithmetic.rs:44
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^256 - 2^224 + 2^192 + 2^96 - 1         p256/src/arithmetic/field.rs:35
    ECGPU_CONST uint32_t P[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u,
                                 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu};
    // n                                            p256/src/lib.rs:60
    ECGPU_CONST uint32_t ORDER[8] = {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu,
                                     0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu};
    // group order in Montgomery form (R = 2^256): R^2 mod n and -n^-1 mod 2^32, for the scalar arithmetic of the
    // ECDSA verification path (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[8] = {0xBE79EEA2u, 0x83244C95u, 0x49BD6FA6u, 0x4699799Cu,
                                           0x2B6BEC59u, 0x2845B239u, 0xF3D95620u, 0x66E12D94u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0xEE00BC4Fu;
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
    ECGPU_CONST int NL = 15;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::P384U;
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
    // group order in Montgomery form (R = 2^384): R^2 mod n and -n^-1 mod 2^32, for the scalar arithmetic of the
    // ECDSA verification path (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[12] = {0x19B409A9u, 0x2D319B24u, 0xDF1AA419u, 0xFF3D81E5u,
                                           0xFCB82947u, 0xBC3E483Au, 0x4AAB1CC5u, 0xD40D4917u,
                                           0x28266895u, 0x3FB05B7Au, 0x2B39BF21u, 0x0C84EE01u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0xE88FDC45u;
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

// SM2 (GB/T 32918.5): the fourth parameter set, SURVEY.md §8(f) rank 4.  Same code path as p256 (a = -3, p = -1 mod 2^64).
struct Sm2Params {
    ECGPU_CONST int ID = CURVE_SM2;
    ECGPU_CONST int N = 8;
    ECGPU_CONST int NL = 10;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::SM2U;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   sm2/src/arithmetic.rs:53-54
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^256 - 2^224 - 2^96 + 2^64 - 1          sm2/src/arithmetic/field.rs:34
    ECGPU_CONST uint32_t P[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu,
                                    0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu};
    // n                                            sm2/src/lib.rs:86
    ECGPU_CONST uint32_t ORDER[8] = {0x39D54123u, 0x53BBF409u, 0x21C6052Bu, 0x7203DF6Bu,
                                    0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu};
    // group order in Montgomery form (R = 2^256): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[8] = {0x7C114F20u, 0x901192AFu, 0xDE6FA2FAu, 0x3464504Au,
                                    0x3AFFE0D4u, 0x620FC84Cu, 0xA22B3D3Bu, 0x1EB5E412u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x72350975u;
    // curve b, canonical                           sm2/src/arithmetic.rs:57-59
    ECGPU_CONST uint32_t B[8] = {0x4D940E93u, 0xDDBCBD41u, 0x15AB8F92u, 0xF39789F5u,
                                    0xCF6509A7u, 0x4D5A9E4Bu, 0x9D9F5E34u, 0x28E9FA9Eu};
    // generator, canonical                         sm2/src/arithmetic.rs:67-74
    ECGPU_CONST uint32_t GX[8] = {0x334C74C7u, 0x715A4589u, 0xF2660BE1u, 0x8FE30BBFu,
                                    0x6A39C994u, 0x5F990446u, 0x1F198119u, 0x32C4AE2Cu};
    ECGPU_CONST uint32_t GY[8] = {0x2139F0A0u, 0x02DF32E5u, 0xC62A4740u, 0xD0A9877Cu,
                                    0x6B692153u, 0x59BDCEE3u, 0xF4F6779Cu, 0xBC3736A2u};
};

// NIST P-224: 7 canonical words (28-byte wire records: 4-byte aligned only), 9 limbs x 27 bits.  p = 1 (mod 4): no
// square root by a single exponentiation, so point decompression is not offered for this curve.
struct P224Params {
    ECGPU_CONST int ID = CURVE_P224;
    ECGPU_CONST int N = 7;
    ECGPU_CONST int NL = 9;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::P224U;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p224/src/arithmetic.rs:41-45
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^224 - 2^96 + 1                          p224/src/arithmetic/field.rs:54-61
    ECGPU_CONST uint32_t P[7] = {0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // n                                            p224/src/lib.rs:50-55
    ECGPU_CONST uint32_t ORDER[7] = {0x5C5C2A3Du, 0x13DD2945u, 0xE0B8F03Eu, 0xFFFF16A2u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order in Montgomery form (R = 2^224): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[7] = {0x3AD01289u, 0x6BDAAE6Cu, 0x97A54552u, 0x6AD09D91u, 0xB1E97961u, 0x1822BC47u, 0xD4BAA4CFu};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x6A1FC2EBu;
    // curve b, canonical                           p224/src/arithmetic.rs:47-50
    ECGPU_CONST uint32_t B[7] = {0x2355FFB4u, 0x270B3943u, 0xD7BFD8BAu, 0x5044B0B7u, 0xF5413256u, 0x0C04B3ABu, 0xB4050A85u};
    // generator, canonical                         p224/src/arithmetic.rs:52-62
    ECGPU_CONST uint32_t GX[7] = {0x115C1D21u, 0x343280D6u, 0x56C21122u, 0x4A03C1D3u, 0x321390B9u, 0x6BB4BF7Fu, 0xB70E0CBDu};
    ECGPU_CONST uint32_t GY[7] = {0x85007E34u, 0x44D58199u, 0x5A074764u, 0xCD4375A0u, 0x4C22DFE6u, 0xB5F723FBu, 0xBD376388u};
};

// NIST P-192: 6 canonical words (24-byte wire records: 8-byte aligned), 8 limbs x 26 bits.
struct P192Params {
    ECGPU_CONST int ID = CURVE_P192;
    ECGPU_CONST int N = 6;
    ECGPU_CONST int NL = 8;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::P192U;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p192/src/arithmetic.rs:39-43
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^192 - 2^64 - 1                          p192/src/arithmetic/field.rs:54
    ECGPU_CONST uint32_t P[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // n                                            p192/src/lib.rs:41
    ECGPU_CONST uint32_t ORDER[6] = {0xB4D22831u, 0x146BC9B1u, 0x99DEF836u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order in Montgomery form (R = 2^192): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[6] = {0xDEB35961u, 0xCE66BACCu, 0xBB3A6BEEu, 0x4696EA5Bu, 0xEA0581A2u, 0x28BE5677u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x0DDBCF2Fu;
    // curve b, canonical                           p192/src/arithmetic.rs:46-47
    ECGPU_CONST uint32_t B[6] = {0xC146B9B1u, 0xFEB8DEECu, 0x72243049u, 0x0FA7E9ABu, 0xE59C80E7u, 0x64210519u};
    // generator, canonical                         p192/src/arithmetic.rs:55-58
    ECGPU_CONST uint32_t GX[6] = {0x82FF1012u, 0xF4FF0AFDu, 0x43A18800u, 0x7CBF20EBu, 0xB03090F6u, 0x188DA80Eu};
    ECGPU_CONST uint32_t GY[6] = {0x1E794811u, 0x73F977A1u, 0x6B24CDD5u, 0x631011EDu, 0xFFC8DA78u, 0x07192B95u};
};

// NIST P-521: 17 canonical words, 66-byte wire records (2-byte aligned: byte-wise access, WireBytes), 20 limbs x 27 bits.
struct P521Params {
    ECGPU_CONST int ID = CURVE_P521;
    ECGPU_CONST int N = 17;
    ECGPU_CONST int NL = 20;
    ECGPU_CONST int WIRE_BYTES = 66;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::P521U;
    ECGPU_CONST bool A_IS_ZERO = false;  // a = -3   p521/src/arithmetic.rs:46,57
    ECGPU_CONST bool MONTGOMERY = true;
    // p = 2^521 - 1                                 p521/src/arithmetic/field.rs:68-80
    ECGPU_CONST uint32_t P[17] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                        0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                        0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000001FFu};
    // n                                            p521/src/lib.rs:51-60
    ECGPU_CONST uint32_t ORDER[17] = {0x91386409u, 0xBB6FB71Eu, 0x899C47AEu, 0x3BB5C9B8u, 0xF709A5D0u, 0x7FCC0148u,
                                        0xBF2F966Bu, 0x51868783u, 0xFFFFFFFAu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                        0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x000001FFu};
    // group order in Montgomery form (R = 2^544): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[17] = {0x61C64CA7u, 0x1163115Au, 0x4374A642u, 0x18354A56u, 0x0791D9DCu, 0x5D4DD6D3u,
                                        0xD3402705u, 0x4FB35B72u, 0xB7756E3Au, 0xCFF3D142u, 0xA8E567BCu, 0x5BCC6D61u,
                                        0x492D0D45u, 0x2D8E03D1u, 0x8C44383Du, 0x5B5A3AFEu, 0x0000019Au};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x79A995C7u;
    // curve b, canonical                           p521/src/arithmetic.rs:62-64
    ECGPU_CONST uint32_t B[17] = {0x6B503F00u, 0xEF451FD4u, 0x3D2C34F1u, 0x3573DF88u, 0x3BB1BF07u, 0x1652C0BDu,
                                        0xEC7E937Bu, 0x56193951u, 0x8EF109E1u, 0xB8B48991u, 0x99B315F3u, 0xA2DA725Bu,
                                        0xB68540EEu, 0x929A21A0u, 0x8E1C9A1Fu, 0x953EB961u, 0x00000051u};
    // generator, canonical                         p521/src/arithmetic.rs:76-83
    ECGPU_CONST uint32_t GX[17] = {0xC2E5BD66u, 0xF97E7E31u, 0x856A429Bu, 0x3348B3C1u, 0xA2FFA8DEu, 0xFE1DC127u,
                                        0xEFE75928u, 0xA14B5E77u, 0x6B4D3DBAu, 0xF828AF60u, 0x053FB521u, 0x9C648139u,
                                        0x2395B442u, 0x9E3ECB66u, 0x0404E9CDu, 0x858E06B7u, 0x000000C6u};
    ECGPU_CONST uint32_t GY[17] = {0x9FD16650u, 0x88BE9476u, 0xA272C240u, 0x353C7086u, 0x3FAD0761u, 0xC550B901u,
                                        0x5EF42640u, 0x97EE7299u, 0x273E662Cu, 0x17AFBD17u, 0x579B4468u, 0x98F54449u,
                                        0x2C7D1BD9u, 0x5C8A5FB4u, 0x9A3BC004u, 0x39296A78u, 0x00000118u};
};

// brainpoolP256r1: a and b generic (RCB Alg 1-3, generic-a Jacobian doubling), general Montgomery constant.
struct Bp256Params {
    ECGPU_CONST int ID = CURVE_BP256;
    ECGPU_CONST int N = 8;
    ECGPU_CONST int NL = 10;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::BP256U;
    ECGPU_CONST bool A_IS_ZERO = false;
    ECGPU_CONST bool A_GENERIC = true;   // bp256/src/r1/arithmetic.rs:35 (EquationAIsGeneric)
    ECGPU_CONST bool MONTGOMERY = true;
    // p                                            bp256/src/arithmetic/field.rs:53
    ECGPU_CONST uint32_t P[8] = {0x1F6E5377u, 0x2013481Du, 0xD5262028u, 0x6E3BF623u, 0x9D838D72u, 0x3E660A90u, 0xA1EEA9BCu, 0xA9FB57DBu};
    // n                                            bp256/src/lib.rs:70
    ECGPU_CONST uint32_t ORDER[8] = {0x974856A7u, 0x901E0E82u, 0xB561A6F7u, 0x8C397AA3u, 0x9D838D71u, 0x3E660A90u, 0xA1EEA9BCu, 0xA9FB57DBu};
    // group order in Montgomery form (R = 2^256): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[8] = {0x3312FCA6u, 0xE1D8D8DEu, 0x1134E4A0u, 0xF35D176Au, 0x6C815CB0u, 0x9B7F25E7u, 0xC3236762u, 0x0B25F1B9u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0xCBB40EE9u;
    // curve b, canonical                           bp256/src/r1/arithmetic.rs:41-43
    ECGPU_CONST uint32_t B[8] = {0xFF8C07B6u, 0x6BCCDC18u, 0x5CF7E1CEu, 0x95841629u, 0xBBD77CBFu, 0xF330B5D9u, 0xE94A4B44u, 0x26DC5C6Cu};
    // generator, canonical                         bp256/src/r1/arithmetic.rs:44-51
    ECGPU_CONST uint32_t GX[8] = {0x9ACE3262u, 0x3A4453BDu, 0xE3BD23C2u, 0xB9DE27E1u, 0xFC81B7AFu, 0x2C4B482Fu, 0xCB7E57CBu, 0x8BD2AEB9u};
    ECGPU_CONST uint32_t GY[8] = {0x2F046997u, 0x5C1D54C7u, 0x2DED8E54u, 0xC2774513u, 0x14611DC9u, 0x97F8461Au, 0xC3DAC4FDu, 0x547EF835u};
};

// brainpoolP384r1: the same generic-a path on 12 words / 15 x 27-bit limbs.
struct Bp384Params {
    ECGPU_CONST int ID = CURVE_BP384;
    ECGPU_CONST int N = 12;
    ECGPU_CONST int NL = 15;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::BP384U;
    ECGPU_CONST bool A_IS_ZERO = false;
    ECGPU_CONST bool A_GENERIC = true;   // bp384/src/r1/arithmetic.rs:33 (EquationAIsGeneric)
    ECGPU_CONST bool MONTGOMERY = true;
    // p                                            bp384/src/arithmetic/field.rs:53
    ECGPU_CONST uint32_t P[12] = {0x3107EC53u, 0x87470013u, 0x901D1A71u, 0xACD3A729u, 0x7FB71123u, 0x12B1DA19u,
                                        0xED5456B4u, 0x152F7109u, 0x50E641DFu, 0x0F5D6F7Eu, 0xA3386D28u, 0x8CB91E82u};
    // n                                            bp384/src/lib.rs:73
    ECGPU_CONST uint32_t ORDER[12] = {0xE9046565u, 0x3B883202u, 0x6B7FC310u, 0xCF3AB6AFu, 0xAC0425A7u, 0x1F166E6Cu,
                                        0xED5456B3u, 0x152F7109u, 0x50E641DFu, 0x0F5D6F7Eu, 0xA3386D28u, 0x8CB91E82u};
    // group order in Montgomery form (R = 2^384): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[12] = {0xDE771C8Eu, 0xAC4ED3A2u, 0x2F2B6B6Eu, 0x37264E20u, 0x9802688Au, 0x2A927E3Bu,
                                        0x52D748FFu, 0x574A74CBu, 0x65165FDBu, 0x8F886DC9u, 0x614E97C2u, 0x0CE8941Au};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x5CB5BB93u;
    // curve b, canonical                           bp384/src/r1/arithmetic.rs:39-41
    ECGPU_CONST uint32_t B[12] = {0xFA504C11u, 0x3AB78696u, 0x95DBC994u, 0x7CB43902u, 0x3EEB62D5u, 0x2E880EA5u,
                                        0x07DCD2A6u, 0x2FB77DE1u, 0x16F0447Cu, 0x8B39B554u, 0x22CE2826u, 0x04A8C7DDu};
    // generator, canonical                         bp384/src/r1/arithmetic.rs:42-49
    ECGPU_CONST uint32_t GX[12] = {0x47D4AF1Eu, 0xEF87B2E2u, 0x36D646AAu, 0xE826E034u, 0x0CBD10E8u, 0xDB7FCAFEu,
                                        0x7EF14FE3u, 0x8847A3E7u, 0xB7C13F6Bu, 0xA2A63A81u, 0x68CF45FFu, 0x1D1C64F0u};
    ECGPU_CONST uint32_t GY[12] = {0x263C5315u, 0x42820341u, 0x77918111u, 0x0E464621u, 0xF9912928u, 0xE19C054Fu,
                                        0xFEEC5864u, 0x62B70B29u, 0x95CFD552u, 0x5CB1EB8Eu, 0x20F9C2A4u, 0x8ABE1D75u};
};

// The brainpool t1 twists: the r1 fields and group orders, their own b and generator, a = -3
// (bp256/src/t1/arithmetic.rs:35, bp384/src/t1/arithmetic.rs:35: EquationAIsMinusThree).  They stay on the any-a code path
// with a = p - 3: the a = -3 formulas need a larger product budget (MAXPROD) than a general p leaves on these limb counts.
// Affine results are the same.
struct Bp256t1Params : Bp256Params {
    ECGPU_CONST int ID = CURVE_BP256T1;
    using UC = consts::BP256T1U;
    // curve b, canonical                           bp256/src/t1/arithmetic.rs:39-41
    ECGPU_CONST uint32_t B[8] = {0xFEE92B04u, 0x6AE58101u, 0xAF2F4925u, 0xBF93EBC4u, 0x3D0B76B7u, 0xFE66A773u, 0x30D84EA4u, 0x662C61C4u};
    // generator, canonical                         bp256/src/t1/arithmetic.rs:42-49
    ECGPU_CONST uint32_t GX[8] = {0x2E1305F4u, 0x79A19156u, 0x7AAFBC2Bu, 0xAFA142C4u, 0x3A656149u, 0x732213B2u, 0xC1CFE7B7u, 0xA3E8EB3Cu};
    ECGPU_CONST uint32_t GY[8] = {0x5B25C9BEu, 0x1DABE8F3u, 0x39D02700u, 0x69BCB6DEu, 0x4644417Eu, 0x7F7B22E1u, 0x3439C56Du, 0x2D996C82u};
};
struct Bp384t1Params : Bp384Params {
    ECGPU_CONST int ID = CURVE_BP384T1;
    using UC = consts::BP384T1U;
    // curve b, canonical                           bp384/src/t1/arithmetic.rs:38-40
    ECGPU_CONST uint32_t B[12] = {0x33B471EEu, 0xED70355Au, 0x3B88805Cu, 0x2074AA26u, 0x756DCE1Du, 0x4B1ABD11u, 0x8CCDC64Eu, 0x4B9346EDu, 0x47910F8Cu, 0xD826DBA6u, 0xA7BDA81Bu, 0x7F519EADu};
    // generator, canonical                         bp384/src/t1/arithmetic.rs:41-48
    ECGPU_CONST uint32_t GX[12] = {0x418808CCu, 0xD8D0AA2Fu, 0x946A5F54u, 0xC4FF191Bu, 0x462AABFFu, 0x2476FECDu, 0xEBD65317u, 0x9B80AB12u, 0x35F72A81u, 0xF2AFCD72u, 0x2DB9A306u, 0x18DE98B0u};
    ECGPU_CONST uint32_t GY[12] = {0x9E582928u, 0x2675BF5Bu, 0x4DC2B291u, 0x46940858u, 0xA208CCFEu, 0x3B88F2B6u, 0x5B7A1FCAu, 0x747F9347u, 0x755AD336u, 0xA114AFD2u, 0x62D30651u, 0x25AB0569u};
};

// bign-curve256v1 (STB 34.101.45-2013; bignp256 in the reference): p = 2^256 - 189, a = -3 — which the reference runs on its
// any-a formulas (bignp256/src/arithmetic.rs:39-40, EquationAIsGeneric), and so does this path with a = p - 3 —, generator
// (0, y).  Its field elements and scalars travel LITTLE-endian (bignp256/src/lib.rs:102 FIELD_ENDIANNESS, arithmetic/field.rs:65,
// arithmetic/scalar.rs:56): WIRE_LE makes every wire accessor read and write the words as they lie.
struct Bign256Params {
    ECGPU_CONST int ID = CURVE_BIGN256;
    ECGPU_CONST int N = 8;
    ECGPU_CONST int NL = 10;
    ECGPU_CONST int REPR = REPR_U28_MONT;
    using UC = consts::BIGN256U;
    ECGPU_CONST bool A_IS_ZERO = false;
    ECGPU_CONST bool A_GENERIC = true;
    ECGPU_CONST bool MONTGOMERY = true;
    ECGPU_CONST bool WIRE_LE = true;
    // p = 2^256 - 189                              bignp256/src/arithmetic/field.rs:60-66
    ECGPU_CONST uint32_t P[8] = {0xFFFFFF43u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // n                                            bignp256/src/lib.rs:74
    ECGPU_CONST uint32_t ORDER[8] = {0x263D6607u, 0x7E5ABF99u, 0x0DFB4DFCu, 0xD95C8ED6u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // group order in Montgomery form (R = 2^256): R^2 mod n and -n^-1 mod 2^32 (ecgpu_scalar.h)
    ECGPU_CONST uint32_t ORDER_R2[8] = {0xDBFF9431u, 0xFA44AF61u, 0x08B44A10u, 0x1B5A5BC1u, 0xA269DBF8u, 0x4A6925C6u, 0xC149A55Bu, 0x05D4EDF1u};
    ECGPU_CONST uint32_t ORDER_NINV32 = 0x0858D849u;
    // curve b, canonical                           bignp256/src/arithmetic.rs:45-47 (little-endian hex there)
    ECGPU_CONST uint32_t B[8] = {0xD69C03F1u, 0xB22E7D6Bu, 0x978B9253u, 0x4CF55069u, 0xE4D8FBBEu, 0xD2C13AABu, 0x15F3A8EDu, 0x77CE6C15u};
    // generator (0, y), canonical                  bignp256/src/arithmetic.rs:48-53
    ECGPU_CONST uint32_t GX[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ECGPU_CONST uint32_t GY[8] = {0x04516A93u, 0x1E29CF18u, 0xC408F652u, 0x78913966u, 0x51D6835Du, 0x5CE4C9A3u, 0xFB16D69Fu, 0x6BF7FC3Cu};
};

// Whether the curve's a is neither 0 nor -3 (the parameter set says A_GENERIC = true)
template <class C, class = void>
struct GenericA {
    static constexpr bool value = false;
};
template <class C>
struct GenericA<C, std::void_t<decltype(C::A_GENERIC)>> {
    static constexpr bool value = C::A_GENERIC;
};

// Wire bytes of a field element / scalar (`FieldBytesSize`): 4 N unless the parameter set says otherwise (p521: 66 bytes
// for 17 words).
template <class C, class = void>
struct WireBytes {
    static constexpr int value = 4 * C::N;
};
template <class C>
struct WireBytes<C, std::void_t<decltype(C::WIRE_BYTES)>> {
    static constexpr int value = C::WIRE_BYTES;
};

// Byte order of the wire records: big-endian (`to_repr` of every curve of the reference but one) unless the parameter set
// says WIRE_LE (bignp256).
template <class C, class = void>
struct WireLe {
    static constexpr bool value = false;
};
template <class C>
struct WireLe<C, std::void_t<decltype(C::WIRE_LE)>> {
    static constexpr bool value = C::WIRE_LE;
};

// one wire record (WireBytes<C> bytes, big-endian unless WireLe<C>) <-> N little-endian words, host or device, any alignment
// the word accessors above accept for 4 N-byte records, byte by byte otherwise
template <class C>
ECGPU_HD void load_be_wire(uint32_t* words, const uint8_t* bytes) {
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WireLe<C>::value) {
        static_assert(WB == 4 * N, "little-endian wire records are whole words");
        for (int i = 0; i < N; i++)
            words[i] = (uint32_t)bytes[4 * i] | (uint32_t)bytes[4 * i + 1] << 8 | (uint32_t)bytes[4 * i + 2] << 16 | (uint32_t)bytes[4 * i + 3] << 24;
    } else if constexpr (WB == 4 * N) {
        load_be<N>(words, bytes);
    } else {
        for (int i = 0; i < N; i++) words[i] = 0;
        for (int j = 0; j < WB; j++) words[(WB - 1 - j) / 4] |= (uint32_t)bytes[j] << (8 * ((WB - 1 - j) % 4));
    }
}
template <class C>
ECGPU_HD void store_be_wire(uint8_t* bytes, const uint32_t* words) {
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WireLe<C>::value) {
        for (int i = 0; i < N; i++)
            for (int j = 0; j < 4; j++) bytes[4 * i + j] = (uint8_t)(words[i] >> (8 * j));
    } else if constexpr (WB == 4 * N) {
        store_be<N>(bytes, words);
    } else {
        for (int j = 0; j < WB; j++) bytes[j] = (uint8_t)(words[(WB - 1 - j) / 4] >> (8 * ((WB - 1 - j) % 4)));
    }
}

}  // namespace ecgpu
