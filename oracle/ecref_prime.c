/*
 * ecref_prime.c — CPU ORACLE (test infrastructure, see ecref.h): NIST P-256 and P-384 field
 * arithmetic restated from the reference, plus two instantiations of the generic primeorder
 * layer (ecref_prime.inc).
 *
 *   p256 field   p256/src/arithmetic/field.rs:59-108, field/field64.rs:7-144
 *                (Montgomery form over U256, hand-written word reduction exploiting p' = 1)
 *   p384 field   p384/src/arithmetic/field.rs:36-57 -> primefield/src/monty.rs:316-382 ->
 *                crypto-bigint 0.7.5 ConstMontyForm (NOT under /root/reference; restated here
 *                as textbook word-by-word Montgomery multiplication, HAC 14.32/14.36)
 */
#include "ecref_internal.h"

/* ======================================================================================
 * Generic Montgomery helpers (crypto-bigint ConstMontyForm semantics), NL words
 * ==================================================================================== */

/* add_mod / sub_mod / neg_mod on fully reduced values */
static void mont_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *p, size_t nl) {
    uint64_t t[9];                                   /* up to nine words (P-521) */
    uint64_t carry = ecref_mp_add(r, a, b, nl);
    uint64_t borrow = ecref_mp_sub(t, r, p, nl);
    if (carry || !borrow) memcpy(r, t, 8 * nl);
}
static void mont_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *p, size_t nl) {
    uint64_t t[9];
    uint64_t borrow = ecref_mp_sub(r, a, b, nl);
    if (borrow) { ecref_mp_add(t, r, p, nl); memcpy(r, t, 8 * nl); }
}

/* Word-by-word Montgomery reduction of a 2*nl-word value t: returns t * R^-1 mod p,
 * m_inv = -p^-1 mod 2^64 (crypto-bigint montgomery_reduction). */
static void mont_reduce(uint64_t *r, const uint64_t *t_in, const uint64_t *p, uint64_t m_inv, size_t nl) {
    uint64_t t[19];
    memcpy(t, t_in, 8 * 2 * nl);
    t[2 * nl] = 0;
    for (size_t i = 0; i < nl; i++) {
        uint64_t u = t[i] * m_inv;
        u128 c = 0;
        for (size_t j = 0; j < nl; j++) {
            c += (u128)u * p[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (size_t k = i + nl; c != 0 && k <= 2 * nl; k++) {
            c += t[k];
            t[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    uint64_t s[9];
    uint64_t borrow = ecref_mp_sub(s, t + nl, p, nl);
    if (t[2 * nl] || !borrow) memcpy(r, s, 8 * nl);
    else memcpy(r, t + nl, 8 * nl);
}

static uint64_t mont_neg_inv64(uint64_t p0) {      /* -p^-1 mod 2^64 by Newton iteration */
    uint64_t x = p0;                               /* correct to 3 bits for odd p0 */
    for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
    return (uint64_t)0 - x;
}

/* R^k mod p by repeated doubling: start from 1 and double 64*nl*k times */
static void mont_pow2_mod(uint64_t *r, const uint64_t *p, size_t nl, size_t doublings) {
    memset(r, 0, 8 * nl);
    r[0] = 1;
    for (size_t i = 0; i < doublings; i++) mont_add(r, r, r, p, nl);
}

/* ======================================================================================
 * P-256 field (Montgomery form)
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe256;

static const uint64_t P256_P[4] = {                     /* p256/src/arithmetic/field.rs:35 */
    0xFFFFFFFFFFFFFFFFULL, 0x00000000FFFFFFFFULL, 0x0000000000000000ULL, 0xFFFFFFFF00000001ULL};
static const uint64_t P256_N[4] = {                     /* p256/src/lib.rs:60 */
    0xF3B9CAC2FC632551ULL, 0xBCE6FAADA7179E84ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
static const uint8_t P256_B_BYTES[32] = {               /* p256/src/arithmetic.rs:55-57 */
    0x5a, 0xc6, 0x35, 0xd8, 0xaa, 0x3a, 0x93, 0xe7, 0xb3, 0xeb, 0xbd, 0x55, 0x76, 0x98, 0x86, 0xbc,
    0x65, 0x1d, 0x06, 0xb0, 0xcc, 0x53, 0xb0, 0xf6, 0x3b, 0xce, 0x3c, 0x3e, 0x27, 0xd2, 0x60, 0x4b};
static const uint8_t P256_GX[32] = {                    /* p256/src/arithmetic.rs:67-74 */
    0x6b, 0x17, 0xd1, 0xf2, 0xe1, 0x2c, 0x42, 0x47, 0xf8, 0xbc, 0xe6, 0xe5, 0x63, 0xa4, 0x40, 0xf2,
    0x77, 0x03, 0x7d, 0x81, 0x2d, 0xeb, 0x33, 0xa0, 0xf4, 0xa1, 0x39, 0x45, 0xd8, 0x98, 0xc2, 0x96};
static const uint8_t P256_GY[32] = {
    0x4f, 0xe3, 0x42, 0xe2, 0xfe, 0x1a, 0x7f, 0x9b, 0x8e, 0xe7, 0xeb, 0x4a, 0x7c, 0x0f, 0x9e, 0x16,
    0x2b, 0xce, 0x33, 0x57, 0x6b, 0x31, 0x5e, 0xce, 0xcb, 0xb6, 0x40, 0x68, 0x37, 0xbf, 0x51, 0xf5};

static fe256 P256_R, P256_R2, P256_B_MONT;
static int p256_ready;

/* carrying_mul_add(a, b, addend, carry) = a*b + addend + carry -> (lo, hi) */
static inline uint64_t cma(uint64_t a, uint64_t b, uint64_t add, uint64_t carry, uint64_t *hi) {
    u128 t = (u128)a * b + add + carry;
    *hi = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
/* carrying_add(a, b, carry) -> (sum, carry) */
static inline uint64_t cadd(uint64_t a, uint64_t b, uint64_t carry, uint64_t *co) {
    u128 t = (u128)a + b + carry;
    *co = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* sub_inner — field64.rs:127-144: (l - r) over five limbs, add p back under the borrow mask */
static void p256_sub_inner(uint64_t out[4], const uint64_t l[5], const uint64_t r[5]) {
    uint64_t w[5], borrow = 0;
    for (int i = 0; i < 5; i++) {
        u128 d = (u128)l[i] - r[i] - borrow;
        w[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    uint64_t mask = (uint64_t)0 - borrow, carry = 0;
    for (int i = 0; i < 4; i++) out[i] = cadd(w[i], P256_P[i] & mask, carry, &carry);
}

/* montgomery_reduce — field64.rs:83-123 */
static fe256 p256_montgomery_reduce(const uint64_t lo[4], const uint64_t hi[4]) {
    uint64_t a0 = lo[0], a1 = lo[1], a2 = lo[2], a3 = lo[3];
    uint64_t a4 = hi[0], a5 = hi[1], a6 = hi[2], a7 = hi[3], a8;
    uint64_t carry, carry2;
    const uint64_t M1 = P256_P[1], M3 = P256_P[3];

    a1 = cma(a0, M1, a1, a0, &carry);
    a2 = cadd(a2, 0, carry, &carry);
    a3 = cma(a0, M3, a3, carry, &carry);
    a4 = cadd(a4, 0, carry, &carry2);

    a2 = cma(a1, M1, a2, a1, &carry);
    a3 = cadd(a3, 0, carry, &carry);
    a4 = cma(a1, M3, a4, carry, &carry);
    a5 = cadd(a5, carry2, carry, &carry2);

    a3 = cma(a2, M1, a3, a2, &carry);
    a4 = cadd(a4, 0, carry, &carry);
    a5 = cma(a2, M3, a5, carry, &carry);
    a6 = cadd(a6, carry2, carry, &carry2);

    a4 = cma(a3, M1, a4, a3, &carry);
    a5 = cadd(a5, 0, carry, &carry);
    a6 = cma(a3, M3, a6, carry, &carry);
    a7 = cadd(a7, carry2, carry, &a8);

    uint64_t l[5] = {a4, a5, a6, a7, a8};
    uint64_t r[5] = {P256_P[0], P256_P[1], P256_P[2], P256_P[3], 0};
    fe256 out;
    p256_sub_inner(out.w, l, r);
    return out;
}

static fe256 p256_fe_mul(const fe256 *a, const fe256 *b) {       /* field.rs:99-102 */
    uint64_t t[8];
    ecref_mp_mul(t, a->w, b->w, 4);                               /* U256::widening_mul */
    return p256_montgomery_reduce(t, t + 4);
}
static fe256 p256_fe_sqr(const fe256 *a) { return p256_fe_mul(a, a); }   /* field.rs:105-108 */

static fe256 p256_fe_add(const fe256 *a, const fe256 *b) {       /* field64.rs:7-23 */
    uint64_t w[5], c = 0;
    for (int i = 0; i < 4; i++) w[i] = cadd(a->w[i], b->w[i], c, &c);
    w[4] = c;
    uint64_t r[5] = {P256_P[0], P256_P[1], P256_P[2], P256_P[3], 0};
    fe256 out;
    p256_sub_inner(out.w, w, r);
    return out;
}
static fe256 p256_fe_sub(const fe256 *a, const fe256 *b) {       /* field64.rs:25-34 */
    uint64_t l[5] = {a->w[0], a->w[1], a->w[2], a->w[3], 0};
    uint64_t r[5] = {b->w[0], b->w[1], b->w[2], b->w[3], 0};
    fe256 out;
    p256_sub_inner(out.w, l, r);
    return out;
}
static fe256 p256_fe_zero(void) { fe256 z = {{0, 0, 0, 0}}; return z; }
static fe256 p256_fe_neg(const fe256 *a) { fe256 z = p256_fe_zero(); return p256_fe_sub(&z, a); }   /* field.rs:74-76 */
static fe256 p256_fe_dbl(const fe256 *a) { return p256_fe_add(a, a); }                                /* field.rs:69-71 */
static int p256_fe_is_zero(const fe256 *a) { return ecref_mp_is_zero(a->w, 4); }

static void p256_init(void) {
    if (p256_ready) return;
    mont_pow2_mod(P256_R.w, P256_P, 4, 256);
    mont_pow2_mod(P256_R2.w, P256_P, 4, 512);                     /* field.rs:183-185 (R2) */
    fe256 b;
    ecref_be_to_words(P256_B_BYTES, 32, b.w);
    P256_B_MONT = p256_fe_mul(&b, &P256_R2);
    p256_ready = 1;
}
static fe256 p256_fe_one(void) { p256_init(); return P256_R; }
static fe256 p256_fe_b(void) { p256_init(); return P256_B_MONT; }

/* from_bytes — field.rs (from_uint: range check, then to Montgomery form via * R2) */
static int p256_fe_from_bytes(fe256 *r, const uint8_t *b) {
    p256_init();
    fe256 t;
    ecref_be_to_words(b, 32, t.w);
    if (ecref_mp_cmp(t.w, P256_P, 4) >= 0) return 0;
    *r = p256_fe_mul(&t, &P256_R2);
    return 1;
}
/* to_bytes — to_canonical = montgomery_reduce(a, 0) field64.rs:37-39 */
static void p256_fe_to_bytes(uint8_t *out, const fe256 *a) {
    uint64_t z[4] = {0, 0, 0, 0};
    fe256 c = p256_montgomery_reduce(a->w, z);
    ecref_words_to_be(c.w, 4, out);
}
/* invert — field.rs (crypto-bigint invert on the Montgomery form); unique result, computed
 * here as a^(p-2) by square-and-multiply. Returns 0 for a == 0. */
static int p256_fe_invert(fe256 *out, const fe256 *a) {
    if (p256_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, P256_P, two, 4);
    fe256 r = p256_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = p256_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p256_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — p256/src/arithmetic/field.rs:121-147: a^((p+1)/4) by the reference's chain, then the root check. */
static fe256 p256_fe_sqn(fe256 a, int n) {
    for (int i = 0; i < n; i++) a = p256_fe_sqr(&a);
    return a;
}
static int p256_fe_sqrt(fe256 *out, const fe256 *a) {
    fe256 t = p256_fe_sqr(a);
    fe256 t11 = p256_fe_mul(a, &t);
    t = p256_fe_sqn(t11, 2);
    fe256 t1111 = p256_fe_mul(&t11, &t);
    t = p256_fe_sqn(t1111, 4);
    fe256 t8 = p256_fe_mul(&t1111, &t);
    t = p256_fe_sqn(t8, 8);
    fe256 x16 = p256_fe_mul(&t, &t8);
    t = p256_fe_sqn(x16, 16);
    t = p256_fe_mul(&t, &x16);
    t = p256_fe_sqn(t, 32);
    t = p256_fe_mul(&t, a);
    t = p256_fe_sqn(t, 96);
    t = p256_fe_mul(&t, a);
    t = p256_fe_sqn(t, 94);
    fe256 sq = p256_fe_sqr(&t);
    fe256 d = p256_fe_sub(&sq, a);
    *out = t;
    return p256_fe_is_zero(&d);
}

#define PO_PFX p256
#define PO_NL 4
#define PO_FE fe256
#define PO_F(name) p256_fe_##name
#define PO_ORDER P256_N
#define PO_GX P256_GX
#define PO_GY P256_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * P-384 field (generic Montgomery, crypto-bigint ConstMontyForm semantics)
 * ==================================================================================== */

typedef struct { uint64_t w[6]; } fe384;

static const uint64_t P384_P[6] = {                     /* p384/src/arithmetic/field.rs:34 */
    0x00000000FFFFFFFFULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFFFFFFFFFEULL,
    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t P384_N[6] = {                     /* p384/src/lib.rs:14 */
    0xECEC196ACCC52973ULL, 0x581A0DB248B0A77AULL, 0xC7634D81F4372DDFULL,
    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint8_t P384_B_BYTES[48] = {               /* p384/src/arithmetic.rs:57-59 */
    0xb3, 0x31, 0x2f, 0xa7, 0xe2, 0x3e, 0xe7, 0xe4, 0x98, 0x8e, 0x05, 0x6b, 0xe3, 0xf8, 0x2d, 0x19,
    0x18, 0x1d, 0x9c, 0x6e, 0xfe, 0x81, 0x41, 0x12, 0x03, 0x14, 0x08, 0x8f, 0x50, 0x13, 0x87, 0x5a,
    0xc6, 0x56, 0x39, 0x8d, 0x8a, 0x2e, 0xd1, 0x9d, 0x2a, 0x85, 0xc8, 0xed, 0xd3, 0xec, 0x2a, 0xef};
static const uint8_t P384_GX[48] = {                    /* p384/src/arithmetic.rs:71-78 */
    0xaa, 0x87, 0xca, 0x22, 0xbe, 0x8b, 0x05, 0x37, 0x8e, 0xb1, 0xc7, 0x1e, 0xf3, 0x20, 0xad, 0x74,
    0x6e, 0x1d, 0x3b, 0x62, 0x8b, 0xa7, 0x9b, 0x98, 0x59, 0xf7, 0x41, 0xe0, 0x82, 0x54, 0x2a, 0x38,
    0x55, 0x02, 0xf2, 0x5d, 0xbf, 0x55, 0x29, 0x6c, 0x3a, 0x54, 0x5e, 0x38, 0x72, 0x76, 0x0a, 0xb7};
static const uint8_t P384_GY[48] = {
    0x36, 0x17, 0xde, 0x4a, 0x96, 0x26, 0x2c, 0x6f, 0x5d, 0x9e, 0x98, 0xbf, 0x92, 0x92, 0xdc, 0x29,
    0xf8, 0xf4, 0x1d, 0xbd, 0x28, 0x9a, 0x14, 0x7c, 0xe9, 0xda, 0x31, 0x13, 0xb5, 0xf0, 0xb8, 0xc0,
    0x0a, 0x60, 0xb1, 0xce, 0x1d, 0x7e, 0x81, 0x9d, 0x7a, 0x43, 0x1d, 0x7c, 0x90, 0xea, 0x0e, 0x5f};

static fe384 P384_R, P384_R2, P384_B_MONT;
static uint64_t P384_MINV;
static int p384_ready;

static fe384 p384_fe_mul(const fe384 *a, const fe384 *b) {       /* monty.rs:346-350 */
    uint64_t t[12];
    fe384 r;
    ecref_mp_mul(t, a->w, b->w, 6);
    mont_reduce(r.w, t, P384_P, P384_MINV, 6);
    return r;
}
static fe384 p384_fe_sqr(const fe384 *a) { return p384_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe384 p384_fe_add(const fe384 *a, const fe384 *b) { fe384 r; mont_add(r.w, a->w, b->w, P384_P, 6); return r; }   /* :316-320 */
static fe384 p384_fe_sub(const fe384 *a, const fe384 *b) { fe384 r; mont_sub(r.w, a->w, b->w, P384_P, 6); return r; }   /* :331-335 */
static fe384 p384_fe_zero(void) { fe384 z; memset(&z, 0, sizeof z); return z; }
static fe384 p384_fe_neg(const fe384 *a) { fe384 z = p384_fe_zero(); return p384_fe_sub(&z, a); }                         /* :353-357 */
static fe384 p384_fe_dbl(const fe384 *a) { return p384_fe_add(a, a); }                                                     /* :323-327 */
static int p384_fe_is_zero(const fe384 *a) { return ecref_mp_is_zero(a->w, 6); }

static void p384_init(void) {
    if (p384_ready) return;
    P384_MINV = mont_neg_inv64(P384_P[0]);
    mont_pow2_mod(P384_R.w, P384_P, 6, 384);
    mont_pow2_mod(P384_R2.w, P384_P, 6, 768);
    fe384 b;
    ecref_be_to_words(P384_B_BYTES, 48, b.w);
    P384_B_MONT = p384_fe_mul(&b, &P384_R2);
    p384_ready = 1;
}
static fe384 p384_fe_one(void) { p384_init(); return P384_R; }
static fe384 p384_fe_b(void) { p384_init(); return P384_B_MONT; }

static int p384_fe_from_bytes(fe384 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    p384_init();
    fe384 t;
    ecref_be_to_words(b, 48, t.w);
    if (ecref_mp_cmp(t.w, P384_P, 6) >= 0) return 0;
    *r = p384_fe_mul(&t, &P384_R2);
    return 1;
}
static void p384_fe_to_bytes(uint8_t *out, const fe384 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[12];
    fe384 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 48);
    mont_reduce(c.w, t, P384_P, P384_MINV, 6);
    ecref_words_to_be(c.w, 6, out);
}
static int p384_fe_invert(fe384 *out, const fe384 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (p384_fe_is_zero(a)) return 0;
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    ecref_mp_sub(e, P384_P, two, 6);
    fe384 r = p384_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = p384_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p384_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int p384_fe_sqrt(fe384 *out, const fe384 *a) {
    uint64_t e[6], one[6] = {1, 0, 0, 0, 0, 0};
    ecref_mp_add(e, P384_P, one, 6);                         /* p + 1 < 2^384 */
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i + 1 < 6 ? e[i + 1] << 62 : 0);
    fe384 r = p384_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = p384_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p384_fe_mul(&r, a);
    }
    fe384 sq = p384_fe_sqr(&r);
    fe384 d = p384_fe_sub(&sq, a);
    *out = r;
    return p384_fe_is_zero(&d);
}

#define PO_PFX p384
#define PO_NL 6
#define PO_FE fe384
#define PO_F(name) p384_fe_##name
#define PO_ORDER P384_N
#define PO_GX P384_GX
#define PO_GY P384_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * SM2 field (generic Montgomery, crypto-bigint ConstMontyForm semantics: sm2/src/arithmetic/field.rs:30-60 ->
 * primefield::MontyFieldElement, like p384) — SURVEY.md 8(f) rank 4
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe_sm2;

static const uint64_t SM2_P[4] = {                      /* sm2/src/arithmetic/field.rs:34 */
    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFEFFFFFFFFULL};
static const uint64_t SM2_N[4] = {                      /* sm2/src/lib.rs:86 */
    0x53BBF40939D54123ULL, 0x7203DF6B21C6052BULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFEFFFFFFFFULL};
static const uint8_t SM2_B_BYTES[32] = {                /* sm2/src/arithmetic.rs:57-59 */
    0x28, 0xe9, 0xfa, 0x9e, 0x9d, 0x9f, 0x5e, 0x34, 0x4d, 0x5a, 0x9e, 0x4b, 0xcf, 0x65, 0x09, 0xa7,
    0xf3, 0x97, 0x89, 0xf5, 0x15, 0xab, 0x8f, 0x92, 0xdd, 0xbc, 0xbd, 0x41, 0x4d, 0x94, 0x0e, 0x93};
static const uint8_t SM2_GX[32] = {                     /* sm2/src/arithmetic.rs:67-74 */
    0x32, 0xc4, 0xae, 0x2c, 0x1f, 0x19, 0x81, 0x19, 0x5f, 0x99, 0x04, 0x46, 0x6a, 0x39, 0xc9, 0x94,
    0x8f, 0xe3, 0x0b, 0xbf, 0xf2, 0x66, 0x0b, 0xe1, 0x71, 0x5a, 0x45, 0x89, 0x33, 0x4c, 0x74, 0xc7};
static const uint8_t SM2_GY[32] = {
    0xbc, 0x37, 0x36, 0xa2, 0xf4, 0xf6, 0x77, 0x9c, 0x59, 0xbd, 0xce, 0xe3, 0x6b, 0x69, 0x21, 0x53,
    0xd0, 0xa9, 0x87, 0x7c, 0xc6, 0x2a, 0x47, 0x40, 0x02, 0xdf, 0x32, 0xe5, 0x21, 0x39, 0xf0, 0xa0};

static fe_sm2 SM2_R, SM2_R2, SM2_B_MONT;
static uint64_t SM2_MINV;
static int sm2_ready;

static fe_sm2 sm2_fe_mul(const fe_sm2 *a, const fe_sm2 *b) {       /* monty.rs:346-350 */
    uint64_t t[8];
    fe_sm2 r;
    ecref_mp_mul(t, a->w, b->w, 4);
    mont_reduce(r.w, t, SM2_P, SM2_MINV, 4);
    return r;
}
static fe_sm2 sm2_fe_sqr(const fe_sm2 *a) { return sm2_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_sm2 sm2_fe_add(const fe_sm2 *a, const fe_sm2 *b) { fe_sm2 r; mont_add(r.w, a->w, b->w, SM2_P, 4); return r; }   /* :316-320 */
static fe_sm2 sm2_fe_sub(const fe_sm2 *a, const fe_sm2 *b) { fe_sm2 r; mont_sub(r.w, a->w, b->w, SM2_P, 4); return r; }   /* :331-335 */
static fe_sm2 sm2_fe_zero(void) { fe_sm2 z; memset(&z, 0, sizeof z); return z; }
static fe_sm2 sm2_fe_neg(const fe_sm2 *a) { fe_sm2 z = sm2_fe_zero(); return sm2_fe_sub(&z, a); }                         /* :353-357 */
static fe_sm2 sm2_fe_dbl(const fe_sm2 *a) { return sm2_fe_add(a, a); }                                                     /* :323-327 */
static int sm2_fe_is_zero(const fe_sm2 *a) { return ecref_mp_is_zero(a->w, 4); }

static void sm2_init(void) {
    if (sm2_ready) return;
    SM2_MINV = mont_neg_inv64(SM2_P[0]);
    mont_pow2_mod(SM2_R.w, SM2_P, 4, 256);
    mont_pow2_mod(SM2_R2.w, SM2_P, 4, 512);
    fe_sm2 b;
    ecref_be_to_words(SM2_B_BYTES, 32, b.w);
    SM2_B_MONT = sm2_fe_mul(&b, &SM2_R2);
    sm2_ready = 1;
}
static fe_sm2 sm2_fe_one(void) { sm2_init(); return SM2_R; }
static fe_sm2 sm2_fe_b(void) { sm2_init(); return SM2_B_MONT; }

static int sm2_fe_from_bytes(fe_sm2 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    sm2_init();
    fe_sm2 t;
    ecref_be_to_words(b, 32, t.w);
    if (ecref_mp_cmp(t.w, SM2_P, 4) >= 0) return 0;
    *r = sm2_fe_mul(&t, &SM2_R2);
    return 1;
}
static void sm2_fe_to_bytes(uint8_t *out, const fe_sm2 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[8];
    fe_sm2 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 32);
    mont_reduce(c.w, t, SM2_P, SM2_MINV, 4);
    ecref_words_to_be(c.w, 4, out);
}
static int sm2_fe_invert(fe_sm2 *out, const fe_sm2 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (sm2_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, SM2_P, two, 4);
    fe_sm2 r = sm2_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = sm2_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = sm2_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int sm2_fe_sqrt(fe_sm2 *out, const fe_sm2 *a) {
    uint64_t e[4], one[4] = {1, 0, 0, 0};
    ecref_mp_add(e, SM2_P, one, 4);                         /* p + 1 < 2^256 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i + 1 < 4 ? e[i + 1] << 62 : 0);
    fe_sm2 r = sm2_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = sm2_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = sm2_fe_mul(&r, a);
    }
    fe_sm2 sq = sm2_fe_sqr(&r);
    fe_sm2 d = sm2_fe_sub(&sq, a);
    *out = r;
    return sm2_fe_is_zero(&d);
}

#define PO_PFX sm2
#define PO_NL 4
#define PO_FE fe_sm2
#define PO_F(name) sm2_fe_##name
#define PO_ORDER SM2_N
#define PO_GX SM2_GX
#define PO_GY SM2_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * P-224 field (generic Montgomery with R = 2^256 on four 64-bit words, crypto-bigint ConstMontyForm semantics:
 * p224/src/arithmetic/field.rs:54-70 -> primefield::MontyFieldElement, like p384; the reference's default backend is the
 * fiat-crypto code for the same Montgomery arithmetic) - SURVEY.md 8(f) rank 4.  Wire elements are 28 bytes.
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe_p224;

static const uint64_t P224_P[4] = {                     /* p224/src/arithmetic/field.rs:54-61 */
    0x0000000000000001ULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFFFFFFFFFFULL, 0x00000000FFFFFFFFULL};
static const uint64_t P224_N[4] = {                     /* p224/src/lib.rs:50-55 */
    0x13DD29455C5C2A3DULL, 0xFFFF16A2E0B8F03EULL, 0xFFFFFFFFFFFFFFFFULL, 0x00000000FFFFFFFFULL};
static const uint8_t P224_B_BYTES[28] = {               /* p224/src/arithmetic.rs:47-50 */
    0xb4, 0x05, 0x0a, 0x85, 0x0c, 0x04, 0xb3, 0xab, 0xf5, 0x41, 0x32, 0x56, 0x50, 0x44, 0xb0, 0xb7, 0xd7, 0xbf, 0xd8, 0xba, 0x27, 0x0b, 0x39, 0x43, 0x23, 0x55, 0xff, 0xb4};
static const uint8_t P224_GX[28] = {                    /* p224/src/arithmetic.rs:52-62 */
    0xb7, 0x0e, 0x0c, 0xbd, 0x6b, 0xb4, 0xbf, 0x7f, 0x32, 0x13, 0x90, 0xb9, 0x4a, 0x03, 0xc1, 0xd3, 0x56, 0xc2, 0x11, 0x22, 0x34, 0x32, 0x80, 0xd6, 0x11, 0x5c, 0x1d, 0x21};
static const uint8_t P224_GY[28] = {
    0xbd, 0x37, 0x63, 0x88, 0xb5, 0xf7, 0x23, 0xfb, 0x4c, 0x22, 0xdf, 0xe6, 0xcd, 0x43, 0x75, 0xa0, 0x5a, 0x07, 0x47, 0x64, 0x44, 0xd5, 0x81, 0x99, 0x85, 0x00, 0x7e, 0x34};

static fe_p224 P224_R, P224_R2, P224_B_MONT;
static uint64_t P224_MINV;
static int p224_ready;

static fe_p224 p224_fe_mul(const fe_p224 *a, const fe_p224 *b) {       /* monty.rs:346-350 */
    uint64_t t[8];
    fe_p224 r;
    ecref_mp_mul(t, a->w, b->w, 4);
    mont_reduce(r.w, t, P224_P, P224_MINV, 4);
    return r;
}
static fe_p224 p224_fe_sqr(const fe_p224 *a) { return p224_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_p224 p224_fe_add(const fe_p224 *a, const fe_p224 *b) { fe_p224 r; mont_add(r.w, a->w, b->w, P224_P, 4); return r; }   /* :316-320 */
static fe_p224 p224_fe_sub(const fe_p224 *a, const fe_p224 *b) { fe_p224 r; mont_sub(r.w, a->w, b->w, P224_P, 4); return r; }   /* :331-335 */
static fe_p224 p224_fe_zero(void) { fe_p224 z; memset(&z, 0, sizeof z); return z; }
static fe_p224 p224_fe_neg(const fe_p224 *a) { fe_p224 z = p224_fe_zero(); return p224_fe_sub(&z, a); }                         /* :353-357 */
static fe_p224 p224_fe_dbl(const fe_p224 *a) { return p224_fe_add(a, a); }                                                     /* :323-327 */
static int p224_fe_is_zero(const fe_p224 *a) { return ecref_mp_is_zero(a->w, 4); }

static void p224_init(void) {
    if (p224_ready) return;
    P224_MINV = mont_neg_inv64(P224_P[0]);
    mont_pow2_mod(P224_R.w, P224_P, 4, 256);
    mont_pow2_mod(P224_R2.w, P224_P, 4, 512);
    fe_p224 b;
    ecref_be_to_words_n(P224_B_BYTES, 28, b.w, 4);
    P224_B_MONT = p224_fe_mul(&b, &P224_R2);
    p224_ready = 1;
}
static fe_p224 p224_fe_one(void) { p224_init(); return P224_R; }
static fe_p224 p224_fe_b(void) { p224_init(); return P224_B_MONT; }

static int p224_fe_from_bytes(fe_p224 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    p224_init();
    fe_p224 t;
    ecref_be_to_words_n(b, 28, t.w, 4);
    if (ecref_mp_cmp(t.w, P224_P, 4) >= 0) return 0;
    *r = p224_fe_mul(&t, &P224_R2);
    return 1;
}
static void p224_fe_to_bytes(uint8_t *out, const fe_p224 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[8];
    fe_p224 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 32);
    mont_reduce(c.w, t, P224_P, P224_MINV, 4);
    ecref_words_to_be_n(c.w, out, 28);
}
static int p224_fe_invert(fe_p224 *out, const fe_p224 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (p224_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, P224_P, two, 4);
    fe_p224 r = p224_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = p224_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p224_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored).  p = 1 (mod 4) with
 * p - 1 = 2^96 * (2^128 - 1): the textbook Tonelli-Shanks loop (Cohen, Algorithm 1.5.1), with z = 11 the smallest
 * quadratic non-residue.  Either root serves: every caller selects by parity (primeorder/src/affine.rs:183-200). */
static int p224_fe_eq(const fe_p224 *a, const fe_p224 *b) { fe_p224 d = p224_fe_sub(a, b); return p224_fe_is_zero(&d); }
static int p224_fe_sqrt(fe_p224 *out, const fe_p224 *a) {
    if (p224_fe_is_zero(a)) { *out = p224_fe_zero(); return 1; }
    const fe_p224 one = p224_fe_one();
    /* q = 2^128 - 1: a^q and a^((q+1)/2) = a^(2^127) by square-and-multiply */
    fe_p224 t = one, r = *a;
    for (int i = 0; i < 128; i++) { t = p224_fe_sqr(&t); t = p224_fe_mul(&t, a); }       /* a^(2^128 - 1) */
    for (int i = 0; i < 127; i++) r = p224_fe_sqr(&r);                                    /* a^(2^127) */
    fe_p224 z = one;
    for (int i = 0; i < 10; i++) z = p224_fe_add(&z, &one);                               /* 11 (Montgomery form) */
    fe_p224 c = one;
    for (int i = 0; i < 128; i++) { c = p224_fe_sqr(&c); c = p224_fe_mul(&c, &z); }       /* z^q: order 2^96 */
    int m = 96;
    while (!p224_fe_eq(&t, &one)) {
        int i = 0;
        fe_p224 u = t;
        while (!p224_fe_eq(&u, &one)) { u = p224_fe_sqr(&u); i++; if (i == m) { *out = p224_fe_zero(); return 0; } }
        fe_p224 b = c;
        for (int j = 0; j < m - i - 1; j++) b = p224_fe_sqr(&b);
        r = p224_fe_mul(&r, &b);
        c = p224_fe_sqr(&b);
        t = p224_fe_mul(&t, &c);
        m = i;
    }
    fe_p224 chk = p224_fe_sqr(&r);
    if (!p224_fe_eq(&chk, a)) { *out = p224_fe_zero(); return 0; }
    *out = r;
    return 1;
}

#define PO_PFX p224
#define PO_NL 4
#define PO_L 28
#define PO_FE fe_p224
#define PO_F(name) p224_fe_##name
#define PO_ORDER P224_N
#define PO_GX P224_GX
#define PO_GY P224_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * P-192 field (generic Montgomery with R = 2^192 on three 64-bit words, crypto-bigint ConstMontyForm semantics:
 * p192/src/arithmetic/field.rs:54-70 -> primefield::MontyFieldElement; the reference's default backend is the
 * fiat-crypto code for the same Montgomery arithmetic) - a parameter set beyond SURVEY.md 8(f) rank 4's list.
 * ==================================================================================== */

typedef struct { uint64_t w[3]; } fe_p192;

static const uint64_t P192_P[3] = {                     /* p192/src/arithmetic/field.rs:54 */
    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t P192_N[3] = {                     /* p192/src/lib.rs:41 */
    0x146BC9B1B4D22831ULL, 0xFFFFFFFF99DEF836ULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint8_t P192_B_BYTES[24] = {               /* p192/src/arithmetic.rs:46-47 */
    0x64, 0x21, 0x05, 0x19, 0xe5, 0x9c, 0x80, 0xe7, 0x0f, 0xa7, 0xe9, 0xab, 0x72, 0x24, 0x30, 0x49, 0xfe, 0xb8, 0xde, 0xec, 0xc1, 0x46, 0xb9, 0xb1};
static const uint8_t P192_GX[24] = {                    /* p192/src/arithmetic.rs:55-58 */
    0x18, 0x8d, 0xa8, 0x0e, 0xb0, 0x30, 0x90, 0xf6, 0x7c, 0xbf, 0x20, 0xeb, 0x43, 0xa1, 0x88, 0x00, 0xf4, 0xff, 0x0a, 0xfd, 0x82, 0xff, 0x10, 0x12};
static const uint8_t P192_GY[24] = {
    0x07, 0x19, 0x2b, 0x95, 0xff, 0xc8, 0xda, 0x78, 0x63, 0x10, 0x11, 0xed, 0x6b, 0x24, 0xcd, 0xd5, 0x73, 0xf9, 0x77, 0xa1, 0x1e, 0x79, 0x48, 0x11};

static fe_p192 P192_R, P192_R2, P192_B_MONT;
static uint64_t P192_MINV;
static int p192_ready;

static fe_p192 p192_fe_mul(const fe_p192 *a, const fe_p192 *b) {       /* monty.rs:346-350 */
    uint64_t t[6];
    fe_p192 r;
    ecref_mp_mul(t, a->w, b->w, 3);
    mont_reduce(r.w, t, P192_P, P192_MINV, 3);
    return r;
}
static fe_p192 p192_fe_sqr(const fe_p192 *a) { return p192_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_p192 p192_fe_add(const fe_p192 *a, const fe_p192 *b) { fe_p192 r; mont_add(r.w, a->w, b->w, P192_P, 3); return r; }   /* :316-320 */
static fe_p192 p192_fe_sub(const fe_p192 *a, const fe_p192 *b) { fe_p192 r; mont_sub(r.w, a->w, b->w, P192_P, 3); return r; }   /* :331-335 */
static fe_p192 p192_fe_zero(void) { fe_p192 z; memset(&z, 0, sizeof z); return z; }
static fe_p192 p192_fe_neg(const fe_p192 *a) { fe_p192 z = p192_fe_zero(); return p192_fe_sub(&z, a); }                         /* :353-357 */
static fe_p192 p192_fe_dbl(const fe_p192 *a) { return p192_fe_add(a, a); }                                                     /* :323-327 */
static int p192_fe_is_zero(const fe_p192 *a) { return ecref_mp_is_zero(a->w, 3); }

static void p192_init(void) {
    if (p192_ready) return;
    P192_MINV = mont_neg_inv64(P192_P[0]);
    mont_pow2_mod(P192_R.w, P192_P, 3, 192);
    mont_pow2_mod(P192_R2.w, P192_P, 3, 384);
    fe_p192 b;
    ecref_be_to_words(P192_B_BYTES, 24, b.w);
    P192_B_MONT = p192_fe_mul(&b, &P192_R2);
    p192_ready = 1;
}
static fe_p192 p192_fe_one(void) { p192_init(); return P192_R; }
static fe_p192 p192_fe_b(void) { p192_init(); return P192_B_MONT; }

static int p192_fe_from_bytes(fe_p192 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    p192_init();
    fe_p192 t;
    ecref_be_to_words(b, 24, t.w);
    if (ecref_mp_cmp(t.w, P192_P, 3) >= 0) return 0;
    *r = p192_fe_mul(&t, &P192_R2);
    return 1;
}
static void p192_fe_to_bytes(uint8_t *out, const fe_p192 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[6];
    fe_p192 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 24);
    mont_reduce(c.w, t, P192_P, P192_MINV, 3);
    ecref_words_to_be(c.w, 3, out);
}
static int p192_fe_invert(fe_p192 *out, const fe_p192 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (p192_fe_is_zero(a)) return 0;
    uint64_t e[3], two[3] = {2, 0, 0};
    ecref_mp_sub(e, P192_P, two, 3);
    fe_p192 r = p192_fe_one();
    for (int i = 191; i >= 0; i--) {
        r = p192_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p192_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int p192_fe_sqrt(fe_p192 *out, const fe_p192 *a) {
    uint64_t e[3], one[3] = {1, 0, 0};
    ecref_mp_add(e, P192_P, one, 3);                         /* p + 1 < 2^192 */
    for (int i = 0; i < 3; i++) e[i] = (e[i] >> 2) | (i + 1 < 3 ? e[i + 1] << 62 : 0);
    fe_p192 r = p192_fe_one();
    for (int i = 191; i >= 0; i--) {
        r = p192_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p192_fe_mul(&r, a);
    }
    fe_p192 sq = p192_fe_sqr(&r);
    fe_p192 d = p192_fe_sub(&sq, a);
    *out = r;
    return p192_fe_is_zero(&d);
}

#define PO_PFX p192
#define PO_NL 3
#define PO_FE fe_p192
#define PO_F(name) p192_fe_##name
#define PO_ORDER P192_N
#define PO_GX P192_GX
#define PO_GY P192_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * P-521 field (generic Montgomery with R = 2^576 on nine 64-bit words, crypto-bigint ConstMontyForm semantics; the
 * reference's own p521 field is a hand-written 9-limb unsaturated backend for the same field, p521/src/arithmetic/field.rs)
 * - SURVEY.md 8(f) rank 4.  Wire elements are 66 bytes.
 * ==================================================================================== */

typedef struct { uint64_t w[9]; } fe_p521;

static const uint64_t P521_P[9] = {                     /* p521/src/arithmetic/field.rs:68-80 */
    0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x00000000000001FFULL};
static const uint64_t P521_N[9] = {                     /* p521/src/lib.rs:51-60 */
    0xBB6FB71E91386409ULL, 0x3BB5C9B8899C47AEULL, 0x7FCC0148F709A5D0ULL, 0x51868783BF2F966BULL, 0xFFFFFFFFFFFFFFFAULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x00000000000001FFULL};
static const uint8_t P521_B_BYTES[66] = {               /* p521/src/arithmetic.rs:62-64 */
    0x00, 0x51, 0x95, 0x3e, 0xb9, 0x61, 0x8e, 0x1c, 0x9a, 0x1f, 0x92, 0x9a, 0x21, 0xa0, 0xb6, 0x85,
    0x40, 0xee, 0xa2, 0xda, 0x72, 0x5b, 0x99, 0xb3, 0x15, 0xf3, 0xb8, 0xb4, 0x89, 0x91, 0x8e, 0xf1,
    0x09, 0xe1, 0x56, 0x19, 0x39, 0x51, 0xec, 0x7e, 0x93, 0x7b, 0x16, 0x52, 0xc0, 0xbd, 0x3b, 0xb1,
    0xbf, 0x07, 0x35, 0x73, 0xdf, 0x88, 0x3d, 0x2c, 0x34, 0xf1, 0xef, 0x45, 0x1f, 0xd4, 0x6b, 0x50,
    0x3f, 0x00};
static const uint8_t P521_GX[66] = {                    /* p521/src/arithmetic.rs:76-83 */
    0x00, 0xc6, 0x85, 0x8e, 0x06, 0xb7, 0x04, 0x04, 0xe9, 0xcd, 0x9e, 0x3e, 0xcb, 0x66, 0x23, 0x95,
    0xb4, 0x42, 0x9c, 0x64, 0x81, 0x39, 0x05, 0x3f, 0xb5, 0x21, 0xf8, 0x28, 0xaf, 0x60, 0x6b, 0x4d,
    0x3d, 0xba, 0xa1, 0x4b, 0x5e, 0x77, 0xef, 0xe7, 0x59, 0x28, 0xfe, 0x1d, 0xc1, 0x27, 0xa2, 0xff,
    0xa8, 0xde, 0x33, 0x48, 0xb3, 0xc1, 0x85, 0x6a, 0x42, 0x9b, 0xf9, 0x7e, 0x7e, 0x31, 0xc2, 0xe5,
    0xbd, 0x66};
static const uint8_t P521_GY[66] = {
    0x01, 0x18, 0x39, 0x29, 0x6a, 0x78, 0x9a, 0x3b, 0xc0, 0x04, 0x5c, 0x8a, 0x5f, 0xb4, 0x2c, 0x7d,
    0x1b, 0xd9, 0x98, 0xf5, 0x44, 0x49, 0x57, 0x9b, 0x44, 0x68, 0x17, 0xaf, 0xbd, 0x17, 0x27, 0x3e,
    0x66, 0x2c, 0x97, 0xee, 0x72, 0x99, 0x5e, 0xf4, 0x26, 0x40, 0xc5, 0x50, 0xb9, 0x01, 0x3f, 0xad,
    0x07, 0x61, 0x35, 0x3c, 0x70, 0x86, 0xa2, 0x72, 0xc2, 0x40, 0x88, 0xbe, 0x94, 0x76, 0x9f, 0xd1,
    0x66, 0x50};

static fe_p521 P521_R, P521_R2, P521_B_MONT;
static uint64_t P521_MINV;
static int p521_ready;

static fe_p521 p521_fe_mul(const fe_p521 *a, const fe_p521 *b) {       /* monty.rs:346-350 */
    uint64_t t[18];
    fe_p521 r;
    ecref_mp_mul(t, a->w, b->w, 9);
    mont_reduce(r.w, t, P521_P, P521_MINV, 9);
    return r;
}
static fe_p521 p521_fe_sqr(const fe_p521 *a) { return p521_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_p521 p521_fe_add(const fe_p521 *a, const fe_p521 *b) { fe_p521 r; mont_add(r.w, a->w, b->w, P521_P, 9); return r; }   /* :316-320 */
static fe_p521 p521_fe_sub(const fe_p521 *a, const fe_p521 *b) { fe_p521 r; mont_sub(r.w, a->w, b->w, P521_P, 9); return r; }   /* :331-335 */
static fe_p521 p521_fe_zero(void) { fe_p521 z; memset(&z, 0, sizeof z); return z; }
static fe_p521 p521_fe_neg(const fe_p521 *a) { fe_p521 z = p521_fe_zero(); return p521_fe_sub(&z, a); }                         /* :353-357 */
static fe_p521 p521_fe_dbl(const fe_p521 *a) { return p521_fe_add(a, a); }                                                     /* :323-327 */
static int p521_fe_is_zero(const fe_p521 *a) { return ecref_mp_is_zero(a->w, 9); }

static void p521_init(void) {
    if (p521_ready) return;
    P521_MINV = mont_neg_inv64(P521_P[0]);
    mont_pow2_mod(P521_R.w, P521_P, 9, 576);
    mont_pow2_mod(P521_R2.w, P521_P, 9, 1152);
    fe_p521 b;
    ecref_be_to_words_n(P521_B_BYTES, 66, b.w, 9);
    P521_B_MONT = p521_fe_mul(&b, &P521_R2);
    p521_ready = 1;
}
static fe_p521 p521_fe_one(void) { p521_init(); return P521_R; }
static fe_p521 p521_fe_b(void) { p521_init(); return P521_B_MONT; }

static int p521_fe_from_bytes(fe_p521 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    p521_init();
    fe_p521 t;
    ecref_be_to_words_n(b, 66, t.w, 9);
    if (ecref_mp_cmp(t.w, P521_P, 9) >= 0) return 0;
    *r = p521_fe_mul(&t, &P521_R2);
    return 1;
}
static void p521_fe_to_bytes(uint8_t *out, const fe_p521 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[18];
    fe_p521 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 72);
    mont_reduce(c.w, t, P521_P, P521_MINV, 9);
    ecref_words_to_be_n(c.w, out, 66);
}
static int p521_fe_invert(fe_p521 *out, const fe_p521 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (p521_fe_is_zero(a)) return 0;
    uint64_t e[9], two[9] = {2, 0, 0, 0, 0, 0, 0, 0, 0};
    ecref_mp_sub(e, P521_P, two, 9);
    fe_p521 r = p521_fe_one();
    for (int i = 575; i >= 0; i--) {
        r = p521_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p521_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int p521_fe_sqrt(fe_p521 *out, const fe_p521 *a) {
    uint64_t e[9], one[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
    ecref_mp_add(e, P521_P, one, 9);                         /* p + 1 = 2^521 < 2^576 */
    for (int i = 0; i < 9; i++) e[i] = (e[i] >> 2) | (i + 1 < 9 ? e[i + 1] << 62 : 0);
    fe_p521 r = p521_fe_one();
    for (int i = 575; i >= 0; i--) {
        r = p521_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = p521_fe_mul(&r, a);
    }
    fe_p521 sq = p521_fe_sqr(&r);
    fe_p521 d = p521_fe_sub(&sq, a);
    *out = r;
    return p521_fe_is_zero(&d);
}

#define PO_PFX p521
#define PO_NL 9
#define PO_L 66
#define PO_FE fe_p521
#define PO_F(name) p521_fe_##name
#define PO_ORDER P521_N
#define PO_GX P521_GX
#define PO_GY P521_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * brainpoolP256r1 field (generic Montgomery, crypto-bigint ConstMontyForm semantics: bp256/src/arithmetic/field.rs:53-62
 * -> primefield::MontyFieldElement) and curve with a generic a (bp256/src/r1/arithmetic.rs:34-52, EquationAIsGeneric)
 * - SURVEY.md 8(f) rank 4.  The reference has no group vectors for this curve: parity rests on the big-int model and on
 * OpenSSL's brainpoolP256r1.
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe_bp256;

static const uint64_t BP256_P[4] = {                    /* bp256/src/arithmetic/field.rs:53 */
    0x2013481D1F6E5377ULL, 0x6E3BF623D5262028ULL, 0x3E660A909D838D72ULL, 0xA9FB57DBA1EEA9BCULL};
static const uint64_t BP256_N[4] = {                    /* bp256/src/lib.rs:70 */
    0x901E0E82974856A7ULL, 0x8C397AA3B561A6F7ULL, 0x3E660A909D838D71ULL, 0xA9FB57DBA1EEA9BCULL};
static const uint8_t BP256_A_BYTES[32] = {              /* bp256/src/r1/arithmetic.rs:38-40 */
    0x7d, 0x5a, 0x09, 0x75, 0xfc, 0x2c, 0x30, 0x57, 0xee, 0xf6, 0x75, 0x30, 0x41, 0x7a, 0xff, 0xe7,
    0xfb, 0x80, 0x55, 0xc1, 0x26, 0xdc, 0x5c, 0x6c, 0xe9, 0x4a, 0x4b, 0x44, 0xf3, 0x30, 0xb5, 0xd9};
static const uint8_t BP256_B_BYTES[32] = {              /* bp256/src/r1/arithmetic.rs:41-43 */
    0x26, 0xdc, 0x5c, 0x6c, 0xe9, 0x4a, 0x4b, 0x44, 0xf3, 0x30, 0xb5, 0xd9, 0xbb, 0xd7, 0x7c, 0xbf,
    0x95, 0x84, 0x16, 0x29, 0x5c, 0xf7, 0xe1, 0xce, 0x6b, 0xcc, 0xdc, 0x18, 0xff, 0x8c, 0x07, 0xb6};
static const uint8_t BP256_GX[32] = {                   /* bp256/src/r1/arithmetic.rs:44-51 */
    0x8b, 0xd2, 0xae, 0xb9, 0xcb, 0x7e, 0x57, 0xcb, 0x2c, 0x4b, 0x48, 0x2f, 0xfc, 0x81, 0xb7, 0xaf,
    0xb9, 0xde, 0x27, 0xe1, 0xe3, 0xbd, 0x23, 0xc2, 0x3a, 0x44, 0x53, 0xbd, 0x9a, 0xce, 0x32, 0x62};
static const uint8_t BP256_GY[32] = {
    0x54, 0x7e, 0xf8, 0x35, 0xc3, 0xda, 0xc4, 0xfd, 0x97, 0xf8, 0x46, 0x1a, 0x14, 0x61, 0x1d, 0xc9,
    0xc2, 0x77, 0x45, 0x13, 0x2d, 0xed, 0x8e, 0x54, 0x5c, 0x1d, 0x54, 0xc7, 0x2f, 0x04, 0x69, 0x97};

static fe_bp256 BP256_R, BP256_R2, BP256_B_MONT, BP256_A_MONT;
static uint64_t BP256_MINV;
static int bp256_ready;

static fe_bp256 bp256_fe_mul(const fe_bp256 *a, const fe_bp256 *b) {       /* monty.rs:346-350 */
    uint64_t t[8];
    fe_bp256 r;
    ecref_mp_mul(t, a->w, b->w, 4);
    mont_reduce(r.w, t, BP256_P, BP256_MINV, 4);
    return r;
}
static fe_bp256 bp256_fe_sqr(const fe_bp256 *a) { return bp256_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_bp256 bp256_fe_add(const fe_bp256 *a, const fe_bp256 *b) { fe_bp256 r; mont_add(r.w, a->w, b->w, BP256_P, 4); return r; }   /* :316-320 */
static fe_bp256 bp256_fe_sub(const fe_bp256 *a, const fe_bp256 *b) { fe_bp256 r; mont_sub(r.w, a->w, b->w, BP256_P, 4); return r; }   /* :331-335 */
static fe_bp256 bp256_fe_zero(void) { fe_bp256 z; memset(&z, 0, sizeof z); return z; }
static fe_bp256 bp256_fe_neg(const fe_bp256 *a) { fe_bp256 z = bp256_fe_zero(); return bp256_fe_sub(&z, a); }                         /* :353-357 */
static fe_bp256 bp256_fe_dbl(const fe_bp256 *a) { return bp256_fe_add(a, a); }                                                     /* :323-327 */
static int bp256_fe_is_zero(const fe_bp256 *a) { return ecref_mp_is_zero(a->w, 4); }

static void bp256_init(void) {
    if (bp256_ready) return;
    BP256_MINV = mont_neg_inv64(BP256_P[0]);
    mont_pow2_mod(BP256_R.w, BP256_P, 4, 256);
    mont_pow2_mod(BP256_R2.w, BP256_P, 4, 512);
    fe_bp256 b;
    ecref_be_to_words(BP256_B_BYTES, 32, b.w);
    BP256_B_MONT = bp256_fe_mul(&b, &BP256_R2);
    ecref_be_to_words(BP256_A_BYTES, 32, b.w);
    BP256_A_MONT = bp256_fe_mul(&b, &BP256_R2);
    bp256_ready = 1;
}
static fe_bp256 bp256_fe_one(void) { bp256_init(); return BP256_R; }
static fe_bp256 bp256_fe_b(void) { bp256_init(); return BP256_B_MONT; }
static fe_bp256 bp256_fe_a(void) { bp256_init(); return BP256_A_MONT; }

static int bp256_fe_from_bytes(fe_bp256 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    bp256_init();
    fe_bp256 t;
    ecref_be_to_words(b, 32, t.w);
    if (ecref_mp_cmp(t.w, BP256_P, 4) >= 0) return 0;
    *r = bp256_fe_mul(&t, &BP256_R2);
    return 1;
}
static void bp256_fe_to_bytes(uint8_t *out, const fe_bp256 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[8];
    fe_bp256 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 32);
    mont_reduce(c.w, t, BP256_P, BP256_MINV, 4);
    ecref_words_to_be(c.w, 4, out);
}
static int bp256_fe_invert(fe_bp256 *out, const fe_bp256 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (bp256_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, BP256_P, two, 4);
    fe_bp256 r = bp256_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bp256_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp256_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int bp256_fe_sqrt(fe_bp256 *out, const fe_bp256 *a) {
    uint64_t e[4], one[4] = {1, 0, 0, 0};
    ecref_mp_add(e, BP256_P, one, 4);                         /* p + 1 < 2^256 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i + 1 < 4 ? e[i + 1] << 62 : 0);
    fe_bp256 r = bp256_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bp256_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp256_fe_mul(&r, a);
    }
    fe_bp256 sq = bp256_fe_sqr(&r);
    fe_bp256 d = bp256_fe_sub(&sq, a);
    *out = r;
    return bp256_fe_is_zero(&d);
}

#define PO_PFX bp256
#define PO_NL 4
#define PO_A_GENERIC 1
#define PO_FE fe_bp256
#define PO_F(name) bp256_fe_##name
#define PO_ORDER BP256_N
#define PO_GX BP256_GX
#define PO_GY BP256_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * bign-curve256v1 (STB 34.101.45; `bignp256`): generic Montgomery field over p = 2^256 - 189 (bignp256/src/arithmetic/field.rs:
 * 60-66 -> primefield::MontyFieldElement) and the curve with a = -3 run on the any-a formulas like the reference
 * (bignp256/src/arithmetic.rs:39-40, EquationAIsGeneric).  This section works on big-endian records like every other one;
 * the curve's LITTLE-endian wire format (bignp256/src/lib.rs:102) is applied at the ABI boundary in ecref.c, which is
 * faithful because the reference's algorithms read `to_be_repr()` / `le_repr()` of the scalar, i.e. its integer value
 * (primeorder/src/tables/radix16.rs:37-38, wnaf/src/lib.rs:197-202).  Pinned by bignp256/src/test_vectors/group.rs.
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe_bign256;

static const uint64_t BIGN256_P[4] = {                  /* bignp256/src/arithmetic/field.rs:60-66: 2^256 - 189 */
    0xFFFFFFFFFFFFFF43ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t BIGN256_N[4] = {                  /* bignp256/src/lib.rs:74 */
    0x7E5ABF99263D6607ULL, 0xD95C8ED60DFB4DFCULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint8_t BIGN256_A_BYTES[32] = {            /* bignp256/src/arithmetic.rs:42-44 (little-endian hex there): p - 3 */
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x40};
static const uint8_t BIGN256_B_BYTES[32] = {            /* bignp256/src/arithmetic.rs:45-47 */
    0x77, 0xce, 0x6c, 0x15, 0x15, 0xf3, 0xa8, 0xed, 0xd2, 0xc1, 0x3a, 0xab, 0xe4, 0xd8, 0xfb, 0xbe,
    0x4c, 0xf5, 0x50, 0x69, 0x97, 0x8b, 0x92, 0x53, 0xb2, 0x2e, 0x7d, 0x6b, 0xd6, 0x9c, 0x03, 0xf1};
static const uint8_t BIGN256_GX[32] = {                 /* bignp256/src/arithmetic.rs:48-53: (0, y) */
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00};
static const uint8_t BIGN256_GY[32] = {
    0x6b, 0xf7, 0xfc, 0x3c, 0xfb, 0x16, 0xd6, 0x9f, 0x5c, 0xe4, 0xc9, 0xa3, 0x51, 0xd6, 0x83, 0x5d,
    0x78, 0x91, 0x39, 0x66, 0xc4, 0x08, 0xf6, 0x52, 0x1e, 0x29, 0xcf, 0x18, 0x04, 0x51, 0x6a, 0x93};

static fe_bign256 BIGN256_R, BIGN256_R2, BIGN256_B_MONT, BIGN256_A_MONT;
static uint64_t BIGN256_MINV;
static int bign256_ready;

static fe_bign256 bign256_fe_mul(const fe_bign256 *a, const fe_bign256 *b) {       /* monty.rs:346-350 */
    uint64_t t[8];
    fe_bign256 r;
    ecref_mp_mul(t, a->w, b->w, 4);
    mont_reduce(r.w, t, BIGN256_P, BIGN256_MINV, 4);
    return r;
}
static fe_bign256 bign256_fe_sqr(const fe_bign256 *a) { return bign256_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_bign256 bign256_fe_add(const fe_bign256 *a, const fe_bign256 *b) { fe_bign256 r; mont_add(r.w, a->w, b->w, BIGN256_P, 4); return r; }   /* :316-320 */
static fe_bign256 bign256_fe_sub(const fe_bign256 *a, const fe_bign256 *b) { fe_bign256 r; mont_sub(r.w, a->w, b->w, BIGN256_P, 4); return r; }   /* :331-335 */
static fe_bign256 bign256_fe_zero(void) { fe_bign256 z; memset(&z, 0, sizeof z); return z; }
static fe_bign256 bign256_fe_neg(const fe_bign256 *a) { fe_bign256 z = bign256_fe_zero(); return bign256_fe_sub(&z, a); }                         /* :353-357 */
static fe_bign256 bign256_fe_dbl(const fe_bign256 *a) { return bign256_fe_add(a, a); }                                                     /* :323-327 */
static int bign256_fe_is_zero(const fe_bign256 *a) { return ecref_mp_is_zero(a->w, 4); }

static void bign256_init(void) {
    if (bign256_ready) return;
    BIGN256_MINV = mont_neg_inv64(BIGN256_P[0]);
    mont_pow2_mod(BIGN256_R.w, BIGN256_P, 4, 256);
    mont_pow2_mod(BIGN256_R2.w, BIGN256_P, 4, 512);
    fe_bign256 b;
    ecref_be_to_words(BIGN256_B_BYTES, 32, b.w);
    BIGN256_B_MONT = bign256_fe_mul(&b, &BIGN256_R2);
    ecref_be_to_words(BIGN256_A_BYTES, 32, b.w);
    BIGN256_A_MONT = bign256_fe_mul(&b, &BIGN256_R2);
    bign256_ready = 1;
}
static fe_bign256 bign256_fe_one(void) { bign256_init(); return BIGN256_R; }
static fe_bign256 bign256_fe_b(void) { bign256_init(); return BIGN256_B_MONT; }
static fe_bign256 bign256_fe_a(void) { bign256_init(); return BIGN256_A_MONT; }

static int bign256_fe_from_bytes(fe_bign256 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    bign256_init();
    fe_bign256 t;
    ecref_be_to_words(b, 32, t.w);
    if (ecref_mp_cmp(t.w, BIGN256_P, 4) >= 0) return 0;
    *r = bign256_fe_mul(&t, &BIGN256_R2);
    return 1;
}
static void bign256_fe_to_bytes(uint8_t *out, const fe_bign256 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[8];
    fe_bign256 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 32);
    mont_reduce(c.w, t, BIGN256_P, BIGN256_MINV, 4);
    ecref_words_to_be(c.w, 4, out);
}
static int bign256_fe_invert(fe_bign256 *out, const fe_bign256 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (bign256_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, BIGN256_P, two, 4);
    fe_bign256 r = bign256_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bign256_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bign256_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int bign256_fe_sqrt(fe_bign256 *out, const fe_bign256 *a) {
    uint64_t e[4], one[4] = {1, 0, 0, 0};
    ecref_mp_add(e, BIGN256_P, one, 4);                         /* p + 1 < 2^256 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i + 1 < 4 ? e[i + 1] << 62 : 0);
    fe_bign256 r = bign256_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bign256_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bign256_fe_mul(&r, a);
    }
    fe_bign256 sq = bign256_fe_sqr(&r);
    fe_bign256 d = bign256_fe_sub(&sq, a);
    *out = r;
    return bign256_fe_is_zero(&d);
}

#define PO_PFX bign256
#define PO_NL 4
#define PO_A_GENERIC 1
#define PO_FE fe_bign256
#define PO_F(name) bign256_fe_##name
#define PO_ORDER BIGN256_N
#define PO_GX BIGN256_GX
#define PO_GY BIGN256_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * brainpoolP384r1: generic Montgomery field on six words (bp384/src/arithmetic/field.rs:53-62) and the generic-a curve
 * (bp384/src/r1/arithmetic.rs:32-50, EquationAIsGeneric).  No group vectors in the reference: parity rests on the big-int
 * model and on OpenSSL's brainpoolP384r1.
 * ==================================================================================== */

typedef struct { uint64_t w[6]; } fe_bp384;

static const uint64_t BP384_P[6] = {                    /* bp384/src/arithmetic/field.rs:53 */
    0x874700133107EC53ULL, 0xACD3A729901D1A71ULL, 0x12B1DA197FB71123ULL, 0x152F7109ED5456B4ULL, 0x0F5D6F7E50E641DFULL, 0x8CB91E82A3386D28ULL};
static const uint64_t BP384_N[6] = {                    /* bp384/src/lib.rs:73 */
    0x3B883202E9046565ULL, 0xCF3AB6AF6B7FC310ULL, 0x1F166E6CAC0425A7ULL, 0x152F7109ED5456B3ULL, 0x0F5D6F7E50E641DFULL, 0x8CB91E82A3386D28ULL};
static const uint8_t BP384_A_BYTES[48] = {              /* bp384/src/r1/arithmetic.rs:36-38 */
    0x7b, 0xc3, 0x82, 0xc6, 0x3d, 0x8c, 0x15, 0x0c, 0x3c, 0x72, 0x08, 0x0a, 0xce, 0x05, 0xaf, 0xa0,
    0xc2, 0xbe, 0xa2, 0x8e, 0x4f, 0xb2, 0x27, 0x87, 0x13, 0x91, 0x65, 0xef, 0xba, 0x91, 0xf9, 0x0f,
    0x8a, 0xa5, 0x81, 0x4a, 0x50, 0x3a, 0xd4, 0xeb, 0x04, 0xa8, 0xc7, 0xdd, 0x22, 0xce, 0x28, 0x26};
static const uint8_t BP384_B_BYTES[48] = {              /* bp384/src/r1/arithmetic.rs:39-41 */
    0x04, 0xa8, 0xc7, 0xdd, 0x22, 0xce, 0x28, 0x26, 0x8b, 0x39, 0xb5, 0x54, 0x16, 0xf0, 0x44, 0x7c,
    0x2f, 0xb7, 0x7d, 0xe1, 0x07, 0xdc, 0xd2, 0xa6, 0x2e, 0x88, 0x0e, 0xa5, 0x3e, 0xeb, 0x62, 0xd5,
    0x7c, 0xb4, 0x39, 0x02, 0x95, 0xdb, 0xc9, 0x94, 0x3a, 0xb7, 0x86, 0x96, 0xfa, 0x50, 0x4c, 0x11};
static const uint8_t BP384_GX[48] = {                   /* bp384/src/r1/arithmetic.rs:42-49 */
    0x1d, 0x1c, 0x64, 0xf0, 0x68, 0xcf, 0x45, 0xff, 0xa2, 0xa6, 0x3a, 0x81, 0xb7, 0xc1, 0x3f, 0x6b,
    0x88, 0x47, 0xa3, 0xe7, 0x7e, 0xf1, 0x4f, 0xe3, 0xdb, 0x7f, 0xca, 0xfe, 0x0c, 0xbd, 0x10, 0xe8,
    0xe8, 0x26, 0xe0, 0x34, 0x36, 0xd6, 0x46, 0xaa, 0xef, 0x87, 0xb2, 0xe2, 0x47, 0xd4, 0xaf, 0x1e};
static const uint8_t BP384_GY[48] = {
    0x8a, 0xbe, 0x1d, 0x75, 0x20, 0xf9, 0xc2, 0xa4, 0x5c, 0xb1, 0xeb, 0x8e, 0x95, 0xcf, 0xd5, 0x52,
    0x62, 0xb7, 0x0b, 0x29, 0xfe, 0xec, 0x58, 0x64, 0xe1, 0x9c, 0x05, 0x4f, 0xf9, 0x91, 0x29, 0x28,
    0x0e, 0x46, 0x46, 0x21, 0x77, 0x91, 0x81, 0x11, 0x42, 0x82, 0x03, 0x41, 0x26, 0x3c, 0x53, 0x15};

static fe_bp384 BP384_R, BP384_R2, BP384_B_MONT, BP384_A_MONT;
static uint64_t BP384_MINV;
static int bp384_ready;

static fe_bp384 bp384_fe_mul(const fe_bp384 *a, const fe_bp384 *b) {       /* monty.rs:346-350 */
    uint64_t t[12];
    fe_bp384 r;
    ecref_mp_mul(t, a->w, b->w, 6);
    mont_reduce(r.w, t, BP384_P, BP384_MINV, 6);
    return r;
}
static fe_bp384 bp384_fe_sqr(const fe_bp384 *a) { return bp384_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_bp384 bp384_fe_add(const fe_bp384 *a, const fe_bp384 *b) { fe_bp384 r; mont_add(r.w, a->w, b->w, BP384_P, 6); return r; }   /* :316-320 */
static fe_bp384 bp384_fe_sub(const fe_bp384 *a, const fe_bp384 *b) { fe_bp384 r; mont_sub(r.w, a->w, b->w, BP384_P, 6); return r; }   /* :331-335 */
static fe_bp384 bp384_fe_zero(void) { fe_bp384 z; memset(&z, 0, sizeof z); return z; }
static fe_bp384 bp384_fe_neg(const fe_bp384 *a) { fe_bp384 z = bp384_fe_zero(); return bp384_fe_sub(&z, a); }                         /* :353-357 */
static fe_bp384 bp384_fe_dbl(const fe_bp384 *a) { return bp384_fe_add(a, a); }                                                     /* :323-327 */
static int bp384_fe_is_zero(const fe_bp384 *a) { return ecref_mp_is_zero(a->w, 6); }

static void bp384_init(void) {
    if (bp384_ready) return;
    BP384_MINV = mont_neg_inv64(BP384_P[0]);
    mont_pow2_mod(BP384_R.w, BP384_P, 6, 384);
    mont_pow2_mod(BP384_R2.w, BP384_P, 6, 768);
    fe_bp384 b;
    ecref_be_to_words(BP384_B_BYTES, 48, b.w);
    BP384_B_MONT = bp384_fe_mul(&b, &BP384_R2);
    ecref_be_to_words(BP384_A_BYTES, 48, b.w);
    BP384_A_MONT = bp384_fe_mul(&b, &BP384_R2);
    bp384_ready = 1;
}
static fe_bp384 bp384_fe_one(void) { bp384_init(); return BP384_R; }
static fe_bp384 bp384_fe_b(void) { bp384_init(); return BP384_B_MONT; }
static fe_bp384 bp384_fe_a(void) { bp384_init(); return BP384_A_MONT; }

static int bp384_fe_from_bytes(fe_bp384 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    bp384_init();
    fe_bp384 t;
    ecref_be_to_words(b, 48, t.w);
    if (ecref_mp_cmp(t.w, BP384_P, 6) >= 0) return 0;
    *r = bp384_fe_mul(&t, &BP384_R2);
    return 1;
}
static void bp384_fe_to_bytes(uint8_t *out, const fe_bp384 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[12];
    fe_bp384 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 48);
    mont_reduce(c.w, t, BP384_P, BP384_MINV, 6);
    ecref_words_to_be(c.w, 6, out);
}
static int bp384_fe_invert(fe_bp384 *out, const fe_bp384 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (bp384_fe_is_zero(a)) return 0;
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    ecref_mp_sub(e, BP384_P, two, 6);
    fe_bp384 r = bp384_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = bp384_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp384_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int bp384_fe_sqrt(fe_bp384 *out, const fe_bp384 *a) {
    uint64_t e[6], one[6] = {1, 0, 0, 0, 0, 0};
    ecref_mp_add(e, BP384_P, one, 6);                         /* p + 1 < 2^384 */
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i + 1 < 6 ? e[i + 1] << 62 : 0);
    fe_bp384 r = bp384_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = bp384_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp384_fe_mul(&r, a);
    }
    fe_bp384 sq = bp384_fe_sqr(&r);
    fe_bp384 d = bp384_fe_sub(&sq, a);
    *out = r;
    return bp384_fe_is_zero(&d);
}

#define PO_PFX bp384
#define PO_NL 6
#define PO_A_GENERIC 1
#define PO_FE fe_bp384
#define PO_F(name) bp384_fe_##name
#define PO_ORDER BP384_N
#define PO_GX BP384_GX
#define PO_GY BP384_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * brainpoolP256t1: the twist of brainpoolP256r1 — same field and group order, a = -3 (bp256/src/t1/arithmetic.rs:35,
 * EquationAIsMinusThree: the a = -3 formulas of ecref_prime.inc), its own b and generator (:38-49).  The field code is repeated
 * under its own names so that the section stands alone.  No group vectors in the reference: parity rests on the big-int model
 * and on OpenSSL's brainpoolP256t1.
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } fe_bp256t1;

static const uint64_t BP256T1_P[4] = {                    /* bp256/src/arithmetic/field.rs:53 */
    0x2013481D1F6E5377ULL, 0x6E3BF623D5262028ULL, 0x3E660A909D838D72ULL, 0xA9FB57DBA1EEA9BCULL};
static const uint64_t BP256T1_N[4] = {                    /* bp256/src/lib.rs:70 */
    0x901E0E82974856A7ULL, 0x8C397AA3B561A6F7ULL, 0x3E660A909D838D71ULL, 0xA9FB57DBA1EEA9BCULL};
static const uint8_t BP256T1_A_BYTES[32] = {
    0xa9, 0xfb, 0x57, 0xdb, 0xa1, 0xee, 0xa9, 0xbc, 0x3e, 0x66, 0x0a, 0x90, 0x9d, 0x83, 0x8d, 0x72,
    0x6e, 0x3b, 0xf6, 0x23, 0xd5, 0x26, 0x20, 0x28, 0x20, 0x13, 0x48, 0x1d, 0x1f, 0x6e, 0x53, 0x74};
static const uint8_t BP256T1_B_BYTES[32] = {
    0x66, 0x2c, 0x61, 0xc4, 0x30, 0xd8, 0x4e, 0xa4, 0xfe, 0x66, 0xa7, 0x73, 0x3d, 0x0b, 0x76, 0xb7,
    0xbf, 0x93, 0xeb, 0xc4, 0xaf, 0x2f, 0x49, 0x25, 0x6a, 0xe5, 0x81, 0x01, 0xfe, 0xe9, 0x2b, 0x04};
static const uint8_t BP256T1_GX[32] = {
    0xa3, 0xe8, 0xeb, 0x3c, 0xc1, 0xcf, 0xe7, 0xb7, 0x73, 0x22, 0x13, 0xb2, 0x3a, 0x65, 0x61, 0x49,
    0xaf, 0xa1, 0x42, 0xc4, 0x7a, 0xaf, 0xbc, 0x2b, 0x79, 0xa1, 0x91, 0x56, 0x2e, 0x13, 0x05, 0xf4};
static const uint8_t BP256T1_GY[32] = {
    0x2d, 0x99, 0x6c, 0x82, 0x34, 0x39, 0xc5, 0x6d, 0x7f, 0x7b, 0x22, 0xe1, 0x46, 0x44, 0x41, 0x7e,
    0x69, 0xbc, 0xb6, 0xde, 0x39, 0xd0, 0x27, 0x00, 0x1d, 0xab, 0xe8, 0xf3, 0x5b, 0x25, 0xc9, 0xbe};

static fe_bp256t1 BP256T1_R, BP256T1_R2, BP256T1_B_MONT, BP256T1_A_MONT;
static uint64_t BP256T1_MINV;
static int bp256t1_ready;

static fe_bp256t1 bp256t1_fe_mul(const fe_bp256t1 *a, const fe_bp256t1 *b) {       /* monty.rs:346-350 */
    uint64_t t[8];
    fe_bp256t1 r;
    ecref_mp_mul(t, a->w, b->w, 4);
    mont_reduce(r.w, t, BP256T1_P, BP256T1_MINV, 4);
    return r;
}
static fe_bp256t1 bp256t1_fe_sqr(const fe_bp256t1 *a) { return bp256t1_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_bp256t1 bp256t1_fe_add(const fe_bp256t1 *a, const fe_bp256t1 *b) { fe_bp256t1 r; mont_add(r.w, a->w, b->w, BP256T1_P, 4); return r; }   /* :316-320 */
static fe_bp256t1 bp256t1_fe_sub(const fe_bp256t1 *a, const fe_bp256t1 *b) { fe_bp256t1 r; mont_sub(r.w, a->w, b->w, BP256T1_P, 4); return r; }   /* :331-335 */
static fe_bp256t1 bp256t1_fe_zero(void) { fe_bp256t1 z; memset(&z, 0, sizeof z); return z; }
static fe_bp256t1 bp256t1_fe_neg(const fe_bp256t1 *a) { fe_bp256t1 z = bp256t1_fe_zero(); return bp256t1_fe_sub(&z, a); }                         /* :353-357 */
static fe_bp256t1 bp256t1_fe_dbl(const fe_bp256t1 *a) { return bp256t1_fe_add(a, a); }                                                     /* :323-327 */
static int bp256t1_fe_is_zero(const fe_bp256t1 *a) { return ecref_mp_is_zero(a->w, 4); }

static void bp256t1_init(void) {
    if (bp256t1_ready) return;
    BP256T1_MINV = mont_neg_inv64(BP256T1_P[0]);
    mont_pow2_mod(BP256T1_R.w, BP256T1_P, 4, 256);
    mont_pow2_mod(BP256T1_R2.w, BP256T1_P, 4, 512);
    fe_bp256t1 b;
    ecref_be_to_words(BP256T1_B_BYTES, 32, b.w);
    BP256T1_B_MONT = bp256t1_fe_mul(&b, &BP256T1_R2);
    ecref_be_to_words(BP256T1_A_BYTES, 32, b.w);
    BP256T1_A_MONT = bp256t1_fe_mul(&b, &BP256T1_R2);
    bp256t1_ready = 1;
}
static fe_bp256t1 bp256t1_fe_one(void) { bp256t1_init(); return BP256T1_R; }
static fe_bp256t1 bp256t1_fe_b(void) { bp256t1_init(); return BP256T1_B_MONT; }
static fe_bp256t1 bp256t1_fe_a(void) { bp256t1_init(); return BP256T1_A_MONT; }

static int bp256t1_fe_from_bytes(fe_bp256t1 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    bp256t1_init();
    fe_bp256t1 t;
    ecref_be_to_words(b, 32, t.w);
    if (ecref_mp_cmp(t.w, BP256T1_P, 4) >= 0) return 0;
    *r = bp256t1_fe_mul(&t, &BP256T1_R2);
    return 1;
}
static void bp256t1_fe_to_bytes(uint8_t *out, const fe_bp256t1 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[8];
    fe_bp256t1 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 32);
    mont_reduce(c.w, t, BP256T1_P, BP256T1_MINV, 4);
    ecref_words_to_be(c.w, 4, out);
}
static int bp256t1_fe_invert(fe_bp256t1 *out, const fe_bp256t1 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (bp256t1_fe_is_zero(a)) return 0;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    ecref_mp_sub(e, BP256T1_P, two, 4);
    fe_bp256t1 r = bp256t1_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bp256t1_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp256t1_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int bp256t1_fe_sqrt(fe_bp256t1 *out, const fe_bp256t1 *a) {
    uint64_t e[4], one[4] = {1, 0, 0, 0};
    ecref_mp_add(e, BP256T1_P, one, 4);                         /* p + 1 < 2^256 */
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i + 1 < 4 ? e[i + 1] << 62 : 0);
    fe_bp256t1 r = bp256t1_fe_one();
    for (int i = 255; i >= 0; i--) {
        r = bp256t1_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp256t1_fe_mul(&r, a);
    }
    fe_bp256t1 sq = bp256t1_fe_sqr(&r);
    fe_bp256t1 d = bp256t1_fe_sub(&sq, a);
    *out = r;
    return bp256t1_fe_is_zero(&d);
}

#define PO_PFX bp256t1
#define PO_NL 4
#define PO_FE fe_bp256t1
#define PO_F(name) bp256t1_fe_##name
#define PO_ORDER BP256T1_N
#define PO_GX BP256T1_GX
#define PO_GY BP256T1_GY
#include "ecref_prime.inc"

/* ======================================================================================
 * brainpoolP384t1: the twist of brainpoolP384r1 — same field and group order, a = -3 (bp384/src/t1/arithmetic.rs:35,
 * EquationAIsMinusThree: the a = -3 formulas of ecref_prime.inc), its own b and generator (:38-49).  The field code is repeated
 * under its own names so that the section stands alone.  No group vectors in the reference: parity rests on the big-int model
 * and on OpenSSL's brainpoolP384t1.
 * ==================================================================================== */

typedef struct { uint64_t w[6]; } fe_bp384t1;

static const uint64_t BP384T1_P[6] = {                    /* bp384/src/arithmetic/field.rs:53 */
    0x874700133107EC53ULL, 0xACD3A729901D1A71ULL, 0x12B1DA197FB71123ULL, 0x152F7109ED5456B4ULL, 0x0F5D6F7E50E641DFULL, 0x8CB91E82A3386D28ULL};
static const uint64_t BP384T1_N[6] = {                    /* bp384/src/lib.rs:73 */
    0x3B883202E9046565ULL, 0xCF3AB6AF6B7FC310ULL, 0x1F166E6CAC0425A7ULL, 0x152F7109ED5456B3ULL, 0x0F5D6F7E50E641DFULL, 0x8CB91E82A3386D28ULL};
static const uint8_t BP384T1_A_BYTES[48] = {
    0x8c, 0xb9, 0x1e, 0x82, 0xa3, 0x38, 0x6d, 0x28, 0x0f, 0x5d, 0x6f, 0x7e, 0x50, 0xe6, 0x41, 0xdf,
    0x15, 0x2f, 0x71, 0x09, 0xed, 0x54, 0x56, 0xb4, 0x12, 0xb1, 0xda, 0x19, 0x7f, 0xb7, 0x11, 0x23,
    0xac, 0xd3, 0xa7, 0x29, 0x90, 0x1d, 0x1a, 0x71, 0x87, 0x47, 0x00, 0x13, 0x31, 0x07, 0xec, 0x50};
static const uint8_t BP384T1_B_BYTES[48] = {
    0x7f, 0x51, 0x9e, 0xad, 0xa7, 0xbd, 0xa8, 0x1b, 0xd8, 0x26, 0xdb, 0xa6, 0x47, 0x91, 0x0f, 0x8c,
    0x4b, 0x93, 0x46, 0xed, 0x8c, 0xcd, 0xc6, 0x4e, 0x4b, 0x1a, 0xbd, 0x11, 0x75, 0x6d, 0xce, 0x1d,
    0x20, 0x74, 0xaa, 0x26, 0x3b, 0x88, 0x80, 0x5c, 0xed, 0x70, 0x35, 0x5a, 0x33, 0xb4, 0x71, 0xee};
static const uint8_t BP384T1_GX[48] = {
    0x18, 0xde, 0x98, 0xb0, 0x2d, 0xb9, 0xa3, 0x06, 0xf2, 0xaf, 0xcd, 0x72, 0x35, 0xf7, 0x2a, 0x81,
    0x9b, 0x80, 0xab, 0x12, 0xeb, 0xd6, 0x53, 0x17, 0x24, 0x76, 0xfe, 0xcd, 0x46, 0x2a, 0xab, 0xff,
    0xc4, 0xff, 0x19, 0x1b, 0x94, 0x6a, 0x5f, 0x54, 0xd8, 0xd0, 0xaa, 0x2f, 0x41, 0x88, 0x08, 0xcc};
static const uint8_t BP384T1_GY[48] = {
    0x25, 0xab, 0x05, 0x69, 0x62, 0xd3, 0x06, 0x51, 0xa1, 0x14, 0xaf, 0xd2, 0x75, 0x5a, 0xd3, 0x36,
    0x74, 0x7f, 0x93, 0x47, 0x5b, 0x7a, 0x1f, 0xca, 0x3b, 0x88, 0xf2, 0xb6, 0xa2, 0x08, 0xcc, 0xfe,
    0x46, 0x94, 0x08, 0x58, 0x4d, 0xc2, 0xb2, 0x91, 0x26, 0x75, 0xbf, 0x5b, 0x9e, 0x58, 0x29, 0x28};

static fe_bp384t1 BP384T1_R, BP384T1_R2, BP384T1_B_MONT, BP384T1_A_MONT;
static uint64_t BP384T1_MINV;
static int bp384t1_ready;

static fe_bp384t1 bp384t1_fe_mul(const fe_bp384t1 *a, const fe_bp384t1 *b) {       /* monty.rs:346-350 */
    uint64_t t[12];
    fe_bp384t1 r;
    ecref_mp_mul(t, a->w, b->w, 6);
    mont_reduce(r.w, t, BP384T1_P, BP384T1_MINV, 6);
    return r;
}
static fe_bp384t1 bp384t1_fe_sqr(const fe_bp384t1 *a) { return bp384t1_fe_mul(a, a); }                /* monty.rs:361-363 */
static fe_bp384t1 bp384t1_fe_add(const fe_bp384t1 *a, const fe_bp384t1 *b) { fe_bp384t1 r; mont_add(r.w, a->w, b->w, BP384T1_P, 6); return r; }   /* :316-320 */
static fe_bp384t1 bp384t1_fe_sub(const fe_bp384t1 *a, const fe_bp384t1 *b) { fe_bp384t1 r; mont_sub(r.w, a->w, b->w, BP384T1_P, 6); return r; }   /* :331-335 */
static fe_bp384t1 bp384t1_fe_zero(void) { fe_bp384t1 z; memset(&z, 0, sizeof z); return z; }
static fe_bp384t1 bp384t1_fe_neg(const fe_bp384t1 *a) { fe_bp384t1 z = bp384t1_fe_zero(); return bp384t1_fe_sub(&z, a); }                         /* :353-357 */
static fe_bp384t1 bp384t1_fe_dbl(const fe_bp384t1 *a) { return bp384t1_fe_add(a, a); }                                                     /* :323-327 */
static int bp384t1_fe_is_zero(const fe_bp384t1 *a) { return ecref_mp_is_zero(a->w, 6); }

static void bp384t1_init(void) {
    if (bp384t1_ready) return;
    BP384T1_MINV = mont_neg_inv64(BP384T1_P[0]);
    mont_pow2_mod(BP384T1_R.w, BP384T1_P, 6, 384);
    mont_pow2_mod(BP384T1_R2.w, BP384T1_P, 6, 768);
    fe_bp384t1 b;
    ecref_be_to_words(BP384T1_B_BYTES, 48, b.w);
    BP384T1_B_MONT = bp384t1_fe_mul(&b, &BP384T1_R2);
    ecref_be_to_words(BP384T1_A_BYTES, 48, b.w);
    BP384T1_A_MONT = bp384t1_fe_mul(&b, &BP384T1_R2);
    bp384t1_ready = 1;
}
static fe_bp384t1 bp384t1_fe_one(void) { bp384t1_init(); return BP384T1_R; }
static fe_bp384t1 bp384t1_fe_b(void) { bp384t1_init(); return BP384T1_B_MONT; }
static fe_bp384t1 bp384t1_fe_a(void) { bp384t1_init(); return BP384T1_A_MONT; }

static int bp384t1_fe_from_bytes(fe_bp384t1 *r, const uint8_t *b) {      /* monty.rs:75-100 */
    bp384t1_init();
    fe_bp384t1 t;
    ecref_be_to_words(b, 48, t.w);
    if (ecref_mp_cmp(t.w, BP384T1_P, 6) >= 0) return 0;
    *r = bp384t1_fe_mul(&t, &BP384T1_R2);
    return 1;
}
static void bp384t1_fe_to_bytes(uint8_t *out, const fe_bp384t1 *a) {     /* monty.rs:249-274 (retrieve) */
    uint64_t t[12];
    fe_bp384t1 c;
    memset(t, 0, sizeof t);
    memcpy(t, a->w, 48);
    mont_reduce(c.w, t, BP384T1_P, BP384T1_MINV, 6);
    ecref_words_to_be(c.w, 6, out);
}
static int bp384t1_fe_invert(fe_bp384t1 *out, const fe_bp384t1 *a) {          /* monty.rs:373-375; a^(p-2) */
    if (bp384t1_fe_is_zero(a)) return 0;
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    ecref_mp_sub(e, BP384T1_P, two, 6);
    fe_bp384t1 r = bp384t1_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = bp384t1_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp384t1_fe_mul(&r, a);
    }
    *out = r;
    return 1;
}

/* sqrt — primefield/src/monty.rs:467-469 -> crypto-bigint ConstMontyForm::sqrt (un-vendored); p = 3 mod 4, so the
 * root is a^((p+1)/4), computed by square-and-multiply, then the root check. */
static int bp384t1_fe_sqrt(fe_bp384t1 *out, const fe_bp384t1 *a) {
    uint64_t e[6], one[6] = {1, 0, 0, 0, 0, 0};
    ecref_mp_add(e, BP384T1_P, one, 6);                         /* p + 1 < 2^384 */
    for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i + 1 < 6 ? e[i + 1] << 62 : 0);
    fe_bp384t1 r = bp384t1_fe_one();
    for (int i = 383; i >= 0; i--) {
        r = bp384t1_fe_sqr(&r);
        if ((e[i / 64] >> (i % 64)) & 1) r = bp384t1_fe_mul(&r, a);
    }
    fe_bp384t1 sq = bp384t1_fe_sqr(&r);
    fe_bp384t1 d = bp384t1_fe_sub(&sq, a);
    *out = r;
    return bp384t1_fe_is_zero(&d);
}

#define PO_PFX bp384t1
#define PO_NL 6
#define PO_FE fe_bp384t1
#define PO_F(name) bp384t1_fe_##name
#define PO_ORDER BP384T1_N
#define PO_GX BP384T1_GX
#define PO_GY BP384T1_GY
#include "ecref_prime.inc"
