/*
 * ecref_k256.c — CPU ORACLE (test infrastructure, see ecref.h): secp256k1 arithmetic restated
 * from RustCrypto/elliptic-curves k256.  Every function cites the reference lines it follows;
 * paths are relative to the reference root.
 *
 *   field        k256/src/arithmetic/field/field_5x52.rs   (5x52-bit lazily reduced limbs)
 *   group law    k256/src/arithmetic/projective.rs:79-247  (RCB-2015 Alg 7/8/9, a=0, b=7)
 *   scalar       k256/src/arithmetic/scalar.rs, scalar/wide64.rs
 *   GLV          k256/src/arithmetic/mul/glv.rs
 *   drivers      k256/src/arithmetic/mul.rs, tables.rs, primeorder/src/tables/{lookup,radix16}.rs
 */
#include "ecref_internal.h"

#include <string.h>

/* ======================================================================================
 * Field: FieldElement5x52
 * ==================================================================================== */

#define M52 0xFFFFFFFFFFFFFULL

/* from_u256_unchecked — field_5x52.rs:64-74 (input: 32 big-endian bytes) */
static fe5 fe5_from_bytes_unchecked(const uint8_t b[32]) {
    uint64_t w[4];
    ecref_be_to_words(b, 32, w);
    fe5 r;
    r.n[0] = w[0] & M52;
    r.n[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    r.n[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    r.n[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    r.n[4] = w[3] >> 16;
    return r;
}

/* get_overflow — field_5x52.rs:110-118 */
static int fe5_get_overflow(const fe5 *a) {
    uint64_t m = a->n[1] & a->n[2] & a->n[3];
    return ((a->n[4] >> 48) != 0) |
           ((a->n[4] == 0x0FFFFFFFFFFFFULL) & (m == M52) & (a->n[0] >= 0xFFFFEFFFFFC2FULL));
}

/* from_bytes — field_5x52.rs:36-38,76-80: returns 0 if the value is >= p */
static int fe5_from_bytes(fe5 *r, const uint8_t b[32]) {
    *r = fe5_from_bytes_unchecked(b);
    return !fe5_get_overflow(r);
}

/* add_modulus_correction — field_5x52.rs:83-101 */
static fe5 fe5_add_modulus_correction(const fe5 *a, uint64_t x) {
    fe5 r;
    uint64_t t0 = a->n[0] + x * 0x1000003D1ULL;
    uint64_t t1 = a->n[1] + (t0 >> 52);
    t0 &= M52;
    uint64_t t2 = a->n[2] + (t1 >> 52);
    t1 &= M52;
    uint64_t t3 = a->n[3] + (t2 >> 52);
    t2 &= M52;
    uint64_t t4 = a->n[4] + (t3 >> 52);
    t3 &= M52;
    r.n[0] = t0; r.n[1] = t1; r.n[2] = t2; r.n[3] = t3; r.n[4] = t4;
    return r;
}

/* normalize_weak — field_5x52.rs:121-133 */
static fe5 fe5_normalize_weak(const fe5 *a) {
    fe5 t = *a;
    uint64_t x = t.n[4] >> 48;        /* subtract_modulus_approximation :104-108 */
    t.n[4] &= 0x0FFFFFFFFFFFFULL;
    return fe5_add_modulus_correction(&t, x);
}

/* normalize — field_5x52.rs:138-155 */
static fe5 fe5_normalize(const fe5 *a) {
    fe5 res = fe5_normalize_weak(a);
    int overflow = fe5_get_overflow(&res);
    fe5 corr = fe5_add_modulus_correction(&res, 1);
    corr.n[4] &= 0x0FFFFFFFFFFFFULL;
    return overflow ? corr : res;
}

/* normalizes_to_zero — field_5x52.rs:158-172 */
static int fe5_normalizes_to_zero(const fe5 *a) {
    fe5 r = fe5_normalize_weak(a);
    uint64_t z0 = r.n[0] | r.n[1] | r.n[2] | r.n[3] | r.n[4];
    uint64_t z1 = (r.n[0] ^ 0x1000003D0ULL) & r.n[1] & r.n[2] & r.n[3] &
                  (r.n[4] ^ 0xF000000000000ULL);
    return (z0 == 0) | (z1 == M52);
}

/* to_bytes — field.rs:110-112 (normalize) + field_5x52.rs:49-61 (to_u256) */
static void fe5_to_bytes(uint8_t out[32], const fe5 *a) {
    fe5 t = fe5_normalize(a);
    uint64_t w[4];
    w[0] = t.n[0] | (t.n[1] << 52);
    w[1] = (t.n[1] >> 12) | (t.n[2] << 40);
    w[2] = (t.n[2] >> 24) | (t.n[3] << 28);
    w[3] = (t.n[3] >> 36) | (t.n[4] << 16);
    ecref_words_to_be(w, 4, out);
}

/* negate(magnitude) — field_5x52.rs:203-211 */
static fe5 fe5_negate(const fe5 *a, uint32_t magnitude) {
    uint64_t m = (uint64_t)magnitude + 1;
    fe5 r;
    r.n[0] = 0xFFFFEFFFFFC2FULL * 2 * m - a->n[0];
    r.n[1] = M52 * 2 * m - a->n[1];
    r.n[2] = M52 * 2 * m - a->n[2];
    r.n[3] = M52 * 2 * m - a->n[3];
    r.n[4] = 0x0FFFFFFFFFFFFULL * 2 * m - a->n[4];
    return r;
}

/* add — field_5x52.rs:215-223 */
static fe5 fe5_add(const fe5 *a, const fe5 *b) {
    fe5 r;
    for (int i = 0; i < 5; i++) r.n[i] = a->n[i] + b->n[i];
    return r;
}

/* double — field.rs:149-151 */
static fe5 fe5_double(const fe5 *a) { return fe5_add(a, a); }

/* mul_single — field_5x52.rs:227-236 */
static fe5 fe5_mul_single(const fe5 *a, uint32_t rhs) {
    fe5 r;
    for (int i = 0; i < 5; i++) r.n[i] = a->n[i] * (uint64_t)rhs;
    return r;
}

/* mul_inner — field_5x52.rs:240-401.
 * Column sums p0..p8 of the 5x5 limb product; the upper columns are folded into the lower
 * ones with R = 2^260 mod p = 0x1000003D10 as they are produced; result magnitude 1. */
static fe5 fe5_mul(const fe5 *x, const fe5 *y) {
    const u128 a0 = x->n[0], a1 = x->n[1], a2 = x->n[2], a3 = x->n[3], a4 = x->n[4];
    const u128 b0 = y->n[0], b1 = y->n[1], b2 = y->n[2], b3 = y->n[3], b4 = y->n[4];
    const u128 m = M52;
    const u128 r = 0x1000003D10ULL;
    u128 c, d;
    uint64_t t3, t4, tx, u0, c64, d64, r0, r1, r2, r3, r4;

    d = a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0;          /* p3 */
    c = a4 * b4;                                          /* p8 */
    d += (c & m) * r;
    c >>= 52;
    c64 = (uint64_t)c;
    t3 = (uint64_t)(d & m);
    d >>= 52;
    d64 = (uint64_t)d;

    d = (u128)d64 + a0 * b4 + a1 * b3 + a2 * b2 + a3 * b1 + a4 * b0;   /* p4 */
    d += (u128)c64 * r;
    t4 = (uint64_t)(d & m);
    d >>= 52;
    d64 = (uint64_t)d;
    tx = t4 >> 48;
    t4 &= (uint64_t)(m >> 4);

    c = a0 * b0;                                          /* p0 */
    d = (u128)d64 + a1 * b4 + a2 * b3 + a3 * b2 + a4 * b1;             /* p5 */
    u0 = (uint64_t)(d & m);
    d >>= 52;
    d64 = (uint64_t)d;
    u0 = (u0 << 4) | tx;
    c += (u128)u0 * (u128)((uint64_t)r >> 4);
    r0 = (uint64_t)(c & m);
    c >>= 52;
    c64 = (uint64_t)c;

    c = (u128)c64 + a0 * b1 + a1 * b0;                    /* p1 */
    d = (u128)d64 + a2 * b4 + a3 * b3 + a4 * b2;          /* p6 */
    c += (d & m) * r;
    d >>= 52;
    d64 = (uint64_t)d;
    r1 = (uint64_t)(c & m);
    c >>= 52;
    c64 = (uint64_t)c;

    c = (u128)c64 + a0 * b2 + a1 * b1 + a2 * b0;          /* p2 */
    d = (u128)d64 + a3 * b4 + a4 * b3;                    /* p7 */
    c += (d & m) * r;
    d >>= 52;
    d64 = (uint64_t)d;
    r2 = (uint64_t)(c & m);
    c >>= 52;
    c64 = (uint64_t)c;

    c = (u128)c64 + (u128)d64 * r + (u128)t3;
    r3 = (uint64_t)(c & m);
    c >>= 52;
    c64 = (uint64_t)c;
    c = (u128)c64 + (u128)t4;
    r4 = (uint64_t)c;

    fe5 out = {{r0, r1, r2, r3, r4}};
    return out;
}

/* square — field_5x52.rs:412-414 (mul_inner(self, self)) */
static fe5 fe5_sqr(const fe5 *a) { return fe5_mul(a, a); }

static fe5 fe5_pow2k(fe5 x, int k) {           /* field.rs:169-175 */
    for (int j = 0; j < k; j++) x = fe5_sqr(&x);
    return x;
}

/* invert — field.rs:178-184.  The reference calls crypto-bigint's safegcd invert_odd_mod on the
 * canonical value; the modular inverse is unique, so the oracle computes a^(p-2) with the usual
 * secp256k1 addition chain (block lengths 223, 22, 1, 2, 1 of p-2) and returns the normalized
 * result (magnitude 1, as the reference documents).  Returns 0 when a == 0. */
static int fe5_invert(fe5 *out, const fe5 *a) {
    if (fe5_normalizes_to_zero(a)) return 0;
    fe5 x = fe5_normalize(a);
    fe5 x2 = fe5_pow2k(x, 1);      x2 = fe5_mul(&x2, &x);
    fe5 x3 = fe5_pow2k(x2, 1);     x3 = fe5_mul(&x3, &x);
    fe5 x6 = fe5_pow2k(x3, 3);     x6 = fe5_mul(&x6, &x3);
    fe5 x9 = fe5_pow2k(x6, 3);     x9 = fe5_mul(&x9, &x3);
    fe5 x11 = fe5_pow2k(x9, 2);    x11 = fe5_mul(&x11, &x2);
    fe5 x22 = fe5_pow2k(x11, 11);  x22 = fe5_mul(&x22, &x11);
    fe5 x44 = fe5_pow2k(x22, 22);  x44 = fe5_mul(&x44, &x22);
    fe5 x88 = fe5_pow2k(x44, 44);  x88 = fe5_mul(&x88, &x44);
    fe5 x176 = fe5_pow2k(x88, 88); x176 = fe5_mul(&x176, &x88);
    fe5 x220 = fe5_pow2k(x176, 44); x220 = fe5_mul(&x220, &x44);
    fe5 x223 = fe5_pow2k(x220, 3); x223 = fe5_mul(&x223, &x3);
    fe5 t = fe5_pow2k(x223, 23);   t = fe5_mul(&t, &x22);
    t = fe5_pow2k(t, 5);           t = fe5_mul(&t, &x);
    t = fe5_pow2k(t, 3);           t = fe5_mul(&t, &x2);
    t = fe5_pow2k(t, 2);           t = fe5_mul(&t, &x);
    *out = fe5_normalize(&t);
    return 1;
}

/* sqrt — field.rs:200-235: a^((p+1)/4) by the reference's addition chain (blocks of 223, 22 and 2 ones), then the
 * is-it-a-root check.  The result has magnitude 1 and is not normalized.  Returns 0 when a is a non-residue. */
static int fe5_sqrt(fe5 *out, const fe5 *a) {
    fe5 x = *a;
    fe5 x2 = fe5_pow2k(x, 1);      x2 = fe5_mul(&x2, &x);
    fe5 x3 = fe5_pow2k(x2, 1);     x3 = fe5_mul(&x3, &x);
    fe5 x6 = fe5_pow2k(x3, 3);     x6 = fe5_mul(&x6, &x3);
    fe5 x9 = fe5_pow2k(x6, 3);     x9 = fe5_mul(&x9, &x3);
    fe5 x11 = fe5_pow2k(x9, 2);    x11 = fe5_mul(&x11, &x2);
    fe5 x22 = fe5_pow2k(x11, 11);  x22 = fe5_mul(&x22, &x11);
    fe5 x44 = fe5_pow2k(x22, 22);  x44 = fe5_mul(&x44, &x22);
    fe5 x88 = fe5_pow2k(x44, 44);  x88 = fe5_mul(&x88, &x44);
    fe5 x176 = fe5_pow2k(x88, 88); x176 = fe5_mul(&x176, &x88);
    fe5 x220 = fe5_pow2k(x176, 44); x220 = fe5_mul(&x220, &x44);
    fe5 x223 = fe5_pow2k(x220, 3); x223 = fe5_mul(&x223, &x3);
    fe5 res = fe5_pow2k(x223, 23); res = fe5_mul(&res, &x22);
    res = fe5_pow2k(res, 6);       res = fe5_mul(&res, &x2);
    res = fe5_pow2k(res, 2);
    fe5 sq = fe5_mul(&res, &res);
    fe5 nsq = fe5_negate(&sq, 1);
    fe5 d = fe5_add(&nsq, a);
    *out = res;
    return fe5_normalizes_to_zero(&d);
}

static const fe5 FE5_ZERO = {{0, 0, 0, 0, 0}};
static const fe5 FE5_ONE = {{1, 0, 0, 0, 0}};

/* ======================================================================================
 * Points: k256 ProjectivePoint / AffinePoint
 * ==================================================================================== */

#define K256_B_SINGLE 7u   /* CURVE_EQUATION_B_SINGLE, k256/src/arithmetic.rs */

typedef struct { fe5 x, y, z; } k256_pt;                 /* projective.rs:40-45 */
typedef struct { fe5 x, y; int infinity; } k256_aff;     /* affine.rs:37-49 */

static k256_pt k256_identity(void) {                     /* projective.rs:49-53 */
    k256_pt p = {FE5_ZERO, FE5_ONE, FE5_ZERO};
    return p;
}

static const uint8_t K256_GX[32] = {                     /* affine.rs:65-79 */
    0x79, 0xbe, 0x66, 0x7e, 0xf9, 0xdc, 0xbb, 0xac, 0x55, 0xa0, 0x62, 0x95, 0xce, 0x87, 0x0b, 0x07,
    0x02, 0x9b, 0xfc, 0xdb, 0x2d, 0xce, 0x28, 0xd9, 0x59, 0xf2, 0x81, 0x5b, 0x16, 0xf8, 0x17, 0x98};
static const uint8_t K256_GY[32] = {
    0x48, 0x3a, 0xda, 0x77, 0x26, 0xa3, 0xc4, 0x65, 0x5d, 0xa4, 0xfb, 0xfc, 0x0e, 0x11, 0x08, 0xa8,
    0xfd, 0x17, 0xb4, 0x48, 0xa6, 0x85, 0x54, 0x19, 0x9c, 0x47, 0xd0, 0x8f, 0xfb, 0x10, 0xd4, 0xb8};
static const uint8_t K256_BETA[32] = {                   /* projective.rs:32-37 */
    0x7a, 0xe9, 0x6a, 0x2b, 0x65, 0x7c, 0x07, 0x10, 0x6e, 0x64, 0x47, 0x9e, 0xac, 0x34, 0x34, 0xe9,
    0x9c, 0xf0, 0x49, 0x75, 0x12, 0xf5, 0x89, 0x95, 0xc1, 0x39, 0x6c, 0x28, 0x71, 0x95, 0x01, 0xee};

static k256_pt k256_generator(void) {                    /* projective.rs:56-60 */
    k256_pt g;
    g.x = fe5_from_bytes_unchecked(K256_GX);
    g.y = fe5_from_bytes_unchecked(K256_GY);
    g.z = FE5_ONE;
    return g;
}

static k256_pt k256_from_affine(const k256_aff *a) {     /* projective.rs From<AffinePoint> */
    if (a->infinity) return k256_identity();
    k256_pt p = {a->x, a->y, FE5_ONE};
    return p;
}

/* neg — projective.rs:79-85 */
static k256_pt k256_neg(const k256_pt *p) {
    k256_pt r = *p;
    fe5 ny = fe5_negate(&p->y, 1);
    r.y = fe5_normalize_weak(&ny);
    return r;
}

/* add_assign — projective.rs:96-131 (RCB Alg 7) */
static k256_pt k256_add(const k256_pt *s, const k256_pt *o) {
    fe5 xx = fe5_mul(&s->x, &o->x);
    fe5 yy = fe5_mul(&s->y, &o->y);
    fe5 zz = fe5_mul(&s->z, &o->z);

    fe5 t, u, v;
    t = fe5_add(&xx, &yy); fe5 n_xx_yy = fe5_negate(&t, 2);
    t = fe5_add(&yy, &zz); fe5 n_yy_zz = fe5_negate(&t, 2);
    t = fe5_add(&xx, &zz); fe5 n_xx_zz = fe5_negate(&t, 2);

    t = fe5_add(&s->x, &s->y); u = fe5_add(&o->x, &o->y); v = fe5_mul(&t, &u);
    fe5 xy_pairs = fe5_add(&v, &n_xx_yy);
    t = fe5_add(&s->y, &s->z); u = fe5_add(&o->y, &o->z); v = fe5_mul(&t, &u);
    fe5 yz_pairs = fe5_add(&v, &n_yy_zz);
    t = fe5_add(&s->x, &s->z); u = fe5_add(&o->x, &o->z); v = fe5_mul(&t, &u);
    fe5 xz_pairs = fe5_add(&v, &n_xx_zz);

    fe5 bzz = fe5_mul_single(&zz, K256_B_SINGLE);
    t = fe5_double(&bzz); t = fe5_add(&t, &bzz);
    fe5 bzz3 = fe5_normalize_weak(&t);

    t = fe5_negate(&bzz3, 1);
    fe5 yy_m_bzz3 = fe5_add(&yy, &t);
    fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);

    t = fe5_mul_single(&yz_pairs, K256_B_SINGLE);
    fe5 byz = fe5_normalize_weak(&t);
    t = fe5_double(&byz); t = fe5_add(&t, &byz);
    fe5 byz3 = fe5_normalize_weak(&t);

    t = fe5_double(&xx);
    fe5 xx3 = fe5_add(&t, &xx);
    t = fe5_double(&xx3); t = fe5_add(&t, &xx3); t = fe5_normalize_weak(&t);
    t = fe5_mul_single(&t, K256_B_SINGLE);
    fe5 bxx9 = fe5_normalize_weak(&t);

    k256_pt r;
    t = fe5_mul(&xy_pairs, &yy_m_bzz3); u = fe5_mul(&byz3, &xz_pairs); u = fe5_negate(&u, 1);
    t = fe5_add(&t, &u); r.x = fe5_normalize_weak(&t);
    t = fe5_mul(&yy_p_bzz3, &yy_m_bzz3); u = fe5_mul(&bxx9, &xz_pairs);
    t = fe5_add(&t, &u); r.y = fe5_normalize_weak(&t);
    t = fe5_mul(&yz_pairs, &yy_p_bzz3); u = fe5_mul(&xx3, &xy_pairs);
    t = fe5_add(&t, &u); r.z = fe5_normalize_weak(&t);
    return r;
}

/* add_assign_mixed — projective.rs:142-176 (RCB Alg 8) */
static k256_pt k256_add_mixed(const k256_pt *s, const k256_aff *o) {
    fe5 t, u, v;
    fe5 xx = fe5_mul(&s->x, &o->x);
    fe5 yy = fe5_mul(&s->y, &o->y);
    t = fe5_add(&s->x, &s->y); u = fe5_add(&o->x, &o->y); v = fe5_mul(&t, &u);
    t = fe5_add(&xx, &yy); t = fe5_negate(&t, 2);
    fe5 xy_pairs = fe5_add(&v, &t);
    t = fe5_mul(&o->y, &s->z); fe5 yz_pairs = fe5_add(&t, &s->y);
    t = fe5_mul(&o->x, &s->z); fe5 xz_pairs = fe5_add(&t, &s->x);

    fe5 bzz = fe5_mul_single(&s->z, K256_B_SINGLE);
    t = fe5_double(&bzz); t = fe5_add(&t, &bzz);
    fe5 bzz3 = fe5_normalize_weak(&t);

    t = fe5_negate(&bzz3, 1);
    fe5 yy_m_bzz3 = fe5_add(&yy, &t);
    fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);

    t = fe5_mul_single(&yz_pairs, K256_B_SINGLE);
    fe5 byz = fe5_normalize_weak(&t);
    t = fe5_double(&byz); t = fe5_add(&t, &byz);
    fe5 byz3 = fe5_normalize_weak(&t);

    t = fe5_double(&xx);
    fe5 xx3 = fe5_add(&t, &xx);
    t = fe5_double(&xx3); t = fe5_add(&t, &xx3); t = fe5_normalize_weak(&t);
    t = fe5_mul_single(&t, K256_B_SINGLE);
    fe5 bxx9 = fe5_normalize_weak(&t);

    k256_pt r;
    t = fe5_mul(&xy_pairs, &yy_m_bzz3); u = fe5_mul(&byz3, &xz_pairs); u = fe5_negate(&u, 1);
    t = fe5_add(&t, &u); r.x = fe5_normalize_weak(&t);
    t = fe5_mul(&yy_p_bzz3, &yy_m_bzz3); u = fe5_mul(&bxx9, &xz_pairs);
    t = fe5_add(&t, &u); r.y = fe5_normalize_weak(&t);
    t = fe5_mul(&yz_pairs, &yy_p_bzz3); u = fe5_mul(&xx3, &xy_pairs);
    t = fe5_add(&t, &u); r.z = fe5_normalize_weak(&t);

    return o->infinity ? *s : r;       /* conditional_assign(.., !other.is_identity()) :173-175 */
}

/* double_in_place — projective.rs:189-217 (RCB Alg 9) */
static k256_pt k256_double(const k256_pt *s) {
    fe5 t, u;
    fe5 yy = fe5_sqr(&s->y);
    fe5 zz = fe5_sqr(&s->z);
    t = fe5_mul(&s->x, &s->y);
    fe5 xy2 = fe5_double(&t);

    fe5 bzz = fe5_mul_single(&zz, K256_B_SINGLE);
    t = fe5_double(&bzz); t = fe5_add(&t, &bzz);
    fe5 bzz3 = fe5_normalize_weak(&t);
    t = fe5_double(&bzz3); t = fe5_add(&t, &bzz3);
    fe5 bzz9 = fe5_normalize_weak(&t);

    t = fe5_negate(&bzz9, 1);
    fe5 yy_m_bzz9 = fe5_add(&yy, &t);
    fe5 yy_p_bzz3 = fe5_add(&yy, &bzz3);

    fe5 yy_zz = fe5_mul(&yy, &zz);
    t = fe5_double(&yy_zz); t = fe5_double(&t);
    fe5 yy_zz8 = fe5_double(&t);
    t = fe5_double(&yy_zz8); t = fe5_add(&t, &yy_zz8); t = fe5_normalize_weak(&t);
    fe5 tt = fe5_mul_single(&t, K256_B_SINGLE);

    k256_pt r;
    r.x = fe5_mul(&xy2, &yy_m_bzz9);
    t = fe5_mul(&yy, &s->y); t = fe5_mul(&t, &s->z);
    t = fe5_double(&t); t = fe5_double(&t); t = fe5_double(&t);
    r.z = fe5_normalize_weak(&t);
    t = fe5_mul(&yy_m_bzz9, &yy_p_bzz3); u = fe5_add(&t, &tt);
    r.y = fe5_normalize_weak(&u);
    return r;
}

/* endomorphism — projective.rs:241-247 */
static k256_pt k256_endomorphism(const k256_pt *p) {
    k256_pt r = *p;
    fe5 beta = fe5_from_bytes_unchecked(K256_BETA);
    r.x = fe5_mul(&p->x, &beta);
    return r;
}

/* to_affine — projective.rs:64-75 */
static k256_aff k256_to_affine(const k256_pt *p) {
    k256_aff a;
    fe5 zinv;
    if (!fe5_invert(&zinv, &p->z)) {
        a.x = FE5_ZERO; a.y = FE5_ZERO; a.infinity = 1;   /* AffinePoint::IDENTITY affine.rs:53-57 */
        return a;
    }
    fe5 x = fe5_mul(&p->x, &zinv), y = fe5_mul(&p->y, &zinv);
    a.x = fe5_normalize(&x); a.y = fe5_normalize(&y); a.infinity = 0;
    return a;
}

/* batch_normalize — projective.rs:367-391 with field.rs:244-265 (Montgomery's trick, zeros skipped) */
static void k256_batch_normalize(const k256_pt *pts, size_t n, k256_aff *out, fe5 *scratch) {
    fe5 acc = FE5_ONE;
    for (size_t i = 0; i < n; i++) {
        scratch[i] = acc;
        if (!fe5_normalizes_to_zero(&pts[i].z)) acc = fe5_mul(&acc, &pts[i].z);
    }
    fe5 inv = acc;
    fe5_invert(&inv, &acc);                    /* acc is a product of non-zero elements */
    acc = inv;
    for (size_t i = n; i-- > 0;) {
        if (fe5_normalizes_to_zero(&pts[i].z)) {
            out[i].x = FE5_ZERO; out[i].y = FE5_ZERO; out[i].infinity = 1;
            continue;
        }
        fe5 t = fe5_mul(&scratch[i], &acc);
        fe5 zinv = fe5_normalize(&t);
        acc = fe5_mul(&acc, &pts[i].z);
        fe5 x = fe5_mul(&pts[i].x, &zinv), y = fe5_mul(&pts[i].y, &zinv);
        out[i].x = fe5_normalize(&x); out[i].y = fe5_normalize(&y); out[i].infinity = 0;
    }
}

/* from_coordinates-style validation: y^2 == x^3 + 7, coordinates < p (affine.rs decoding) */
static int k256_aff_from_bytes(k256_aff *a, const uint8_t *xy, int inf) {
    if (inf) { a->x = FE5_ZERO; a->y = FE5_ZERO; a->infinity = 1; return 1; }
    if (!fe5_from_bytes(&a->x, xy) || !fe5_from_bytes(&a->y, xy + 32)) return 0;
    a->infinity = 0;
    fe5 lhs = fe5_sqr(&a->y);
    fe5 x2 = fe5_sqr(&a->x), x3 = fe5_mul(&x2, &a->x);
    fe5 seven = {{7, 0, 0, 0, 0}};
    fe5 rhs = fe5_add(&x3, &seven);
    fe5 nl = fe5_negate(&lhs, 1);
    fe5 diff = fe5_add(&rhs, &nl);
    return fe5_normalizes_to_zero(&diff);
}

static void k256_aff_to_bytes(const k256_aff *a, uint8_t *xy, uint8_t *inf) {
    if (a->infinity) { memset(xy, 0, 64); if (inf) *inf = 1; return; }
    fe5_to_bytes(xy, &a->x); fe5_to_bytes(xy + 32, &a->y);
    if (inf) *inf = 0;
}

/* ======================================================================================
 * Scalars: k256 Scalar (canonical U256, little-endian 64-bit words) and WideScalar
 * ==================================================================================== */

typedef struct { uint64_t w[4]; } sc4;

static const uint64_t K256_N[4] = {                       /* k256/src/lib.rs:71 (ORDER) */
    0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t K256_NEG_N[4] = {                   /* wide64.rs:11 NEG_MODULUS = 2^256 - n */
    0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1, 0};
static const uint64_t K256_FRAC_N_2[4] = {                /* scalar.rs FRAC_MODULUS_2 = n >> 1 */
    0xDFE92F46681B20A0ULL, 0x5D576E7357A4501DULL, 0xFFFFFFFFFFFFFFFFULL, 0x7FFFFFFFFFFFFFFFULL};

static sc4 sc4_from_be(const uint8_t b[32]) { sc4 r; ecref_be_to_words(b, 32, r.w); return r; }
static void sc4_to_be(const sc4 *a, uint8_t b[32]) { ecref_words_to_be(a->w, 4, b); }

/* Scalar::add — scalar.rs:106-108 (add_mod) */
static sc4 sc4_add(const sc4 *a, const sc4 *b) {
    sc4 r, t;
    uint64_t carry = ecref_mp_add(r.w, a->w, b->w, 4);
    uint64_t borrow = ecref_mp_sub(t.w, r.w, K256_N, 4);
    if (carry || !borrow) return t;
    return r;
}

/* Scalar::negate — scalar.rs:100-102 (neg_mod) */
static sc4 sc4_neg(const sc4 *a) {
    sc4 r;
    if ((a->w[0] | a->w[1] | a->w[2] | a->w[3]) == 0) return *a;
    ecref_mp_sub(r.w, K256_N, a->w, 4);
    return r;
}

/* IsHigh — scalar.rs:419-423: a > n/2 */
static int sc4_is_high(const sc4 *a) { return ecref_mp_cmp(a->w, K256_FRAC_N_2, 4) > 0; }

/* WideScalar::mul_wide — wide64.rs:23-59 (schoolbook 4x4 -> 8 words) */
static void sc4_mul_wide(uint64_t l[8], const sc4 *a, const sc4 *b) {
    ecref_mp_mul(l, a->w, b->w, 4);
}

/* WideScalar::reduce_impl(false) — wide64.rs:121-212: fold the high words with 2^256 - n
 * three times (512 -> 385 -> 258 -> 256 bits) and do one conditional subtraction. */
static sc4 sc4_reduce_wide(const uint64_t l[8]) {
    uint64_t m[7], p[5], t[8];
    /* m = l[0..3] + l[4..7] * NEG_N  (NEG_N has 129 bits -> product 385 bits) */
    uint64_t prod[7] = {0};
    {
        uint64_t hi[4] = {l[4], l[5], l[6], l[7]};
        uint64_t neg3[3] = {K256_NEG_N[0], K256_NEG_N[1], K256_NEG_N[2]};
        uint64_t full[7];
        ecref_mp_mul_rect(full, hi, 4, neg3, 3);
        memcpy(prod, full, sizeof(prod));
    }
    uint64_t lo7[7] = {l[0], l[1], l[2], l[3], 0, 0, 0};
    ecref_mp_add(m, prod, lo7, 7);
    /* p = m[0..3] + m[4..6] * NEG_N  (258 bits) */
    {
        uint64_t hi[3] = {m[4], m[5], m[6]};
        uint64_t neg3[3] = {K256_NEG_N[0], K256_NEG_N[1], K256_NEG_N[2]};
        uint64_t full[6];
        ecref_mp_mul_rect(full, hi, 3, neg3, 3);
        uint64_t lo5[5] = {m[0], m[1], m[2], m[3], 0};
        uint64_t f5[5] = {full[0], full[1], full[2], full[3], full[4]};
        ecref_mp_add(p, f5, lo5, 5);
    }
    /* r = p[0..3] + p[4] * NEG_N, then final conditional subtraction of n */
    {
        uint64_t hi[1] = {p[4]};
        uint64_t neg3[3] = {K256_NEG_N[0], K256_NEG_N[1], K256_NEG_N[2]};
        uint64_t full[4];
        ecref_mp_mul_rect(full, neg3, 3, hi, 1);
        uint64_t lo5[5] = {p[0], p[1], p[2], p[3], 0};
        uint64_t f5[5] = {full[0], full[1], full[2], full[3], 0};
        ecref_mp_add(t, f5, lo5, 5);
    }
    sc4 r = {{t[0], t[1], t[2], t[3]}}, s;
    uint64_t borrow = ecref_mp_sub(s.w, r.w, K256_N, 4);
    if (t[4] || !borrow) return s;
    return r;
}

/* Scalar::mul — scalar.rs:120-122 */
static sc4 sc4_mul(const sc4 *a, const sc4 *b) {
    uint64_t l[8];
    sc4_mul_wide(l, a, b);
    return sc4_reduce_wide(l);
}

/* WideScalar::mul_shift_vartime(a, b, 384) — wide64.rs:64-119: floor(a*b / 2^384) rounded to
 * nearest by adding bit 383 of the product. */
static sc4 sc4_mul_shift_384(const sc4 *a, const sc4 *b) {
    uint64_t l[8];
    sc4_mul_wide(l, a, b);
    sc4 res = {{l[6], l[7], 0, 0}};
    uint64_t c = (l[5] >> 63) & 1;
    if (c) {
        sc4 one = {{1, 0, 0, 0}};
        res = sc4_add(&res, &one);
    }
    return res;
}

static const uint8_t K256_MINUS_LAMBDA[32] = {            /* glv.rs:10-13 */
    0xac, 0x9c, 0x52, 0xb3, 0x3f, 0xa3, 0xcf, 0x1f, 0x5a, 0xd9, 0xe3, 0xfd, 0x77, 0xed, 0x9b, 0xa4,
    0xa8, 0x80, 0xb9, 0xfc, 0x8e, 0xc7, 0x39, 0xc2, 0xe0, 0xcf, 0xc8, 0x10, 0xb5, 0x12, 0x83, 0xcf};
static const uint8_t K256_MINUS_B1[32] = {                /* glv.rs:16-19 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0xe4, 0x43, 0x7e, 0xd6, 0x01, 0x0e, 0x88, 0x28, 0x6f, 0x54, 0x7f, 0xa9, 0x0a, 0xbf, 0xe4, 0xc3};
static const uint8_t K256_MINUS_B2[32] = {                /* glv.rs:22-25 */
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xfe,
    0x8a, 0x28, 0x0a, 0xc5, 0x07, 0x74, 0x34, 0x6d, 0xd7, 0x65, 0xcd, 0xa8, 0x3d, 0xb1, 0x56, 0x2c};
static const uint8_t K256_G1[32] = {                      /* glv.rs:28-31 */
    0x30, 0x86, 0xd2, 0x21, 0xa7, 0xd4, 0x6b, 0xcd, 0xe8, 0x6c, 0x90, 0xe4, 0x92, 0x84, 0xeb, 0x15,
    0x3d, 0xaa, 0x8a, 0x14, 0x71, 0xe8, 0xca, 0x7f, 0xe8, 0x93, 0x20, 0x9a, 0x45, 0xdb, 0xb0, 0x31};
static const uint8_t K256_G2[32] = {                      /* glv.rs:34-37 */
    0xe4, 0x43, 0x7e, 0xd6, 0x01, 0x0e, 0x88, 0x28, 0x6f, 0x54, 0x7f, 0xa9, 0x0a, 0xbf, 0xe4, 0xc4,
    0x22, 0x12, 0x08, 0xac, 0x9d, 0xf5, 0x06, 0xc6, 0x15, 0x71, 0xb4, 0xae, 0x8a, 0xc4, 0x7f, 0x71};

/* glv::decompose_scalar — mul/glv.rs:149-156 */
static void k256_decompose_scalar(const sc4 *k, sc4 *r1, sc4 *r2) {
    sc4 g1 = sc4_from_be(K256_G1), g2 = sc4_from_be(K256_G2);
    sc4 mb1 = sc4_from_be(K256_MINUS_B1), mb2 = sc4_from_be(K256_MINUS_B2);
    sc4 ml = sc4_from_be(K256_MINUS_LAMBDA);
    sc4 t = sc4_mul_shift_384(k, &g1);
    sc4 c1 = sc4_mul(&t, &mb1);
    t = sc4_mul_shift_384(k, &g2);
    sc4 c2 = sc4_mul(&t, &mb2);
    *r2 = sc4_add(&c1, &c2);
    t = sc4_mul(r2, &ml);
    *r1 = sc4_add(k, &t);
}

int ecref_k256_glv_decompose(const uint8_t *k, uint8_t *r1, uint8_t *r2) {
    sc4 kk = sc4_from_be(k), a, b;
    if (ecref_mp_cmp(kk.w, K256_N, 4) >= 0) return ECREF_ERR_SCALAR_RANGE;
    k256_decompose_scalar(&kk, &a, &b);
    sc4_to_be(&a, r1);
    sc4_to_be(&b, r2);
    return ECREF_OK;
}

/* ======================================================================================
 * LookupTable (primeorder/src/tables/lookup.rs:18-82) specialised to k256 points
 * ==================================================================================== */

typedef struct { k256_pt points[8]; } k256_lut;

static void k256_lut_new(k256_lut *t, const k256_pt *p) {          /* lookup.rs:30-38 */
    t->points[0] = *p;
    for (int j = 0; j < 7; j++) t->points[j + 1] = k256_add(p, &t->points[j]);
}

static k256_pt k256_lut_select(const k256_lut *t, int8_t x) {      /* lookup.rs:43-65 */
    int8_t xmask = (int8_t)(x >> 7);
    int8_t xabs = (int8_t)((x + xmask) ^ xmask);
    k256_pt r = k256_identity();
    for (int j = 1; j <= 8; j++)
        if (xabs == j) r = t->points[j - 1];
    if (xmask & 1) r = k256_neg(&r);
    return r;
}

/* ======================================================================================
 * Drivers — k256/src/arithmetic/mul.rs
 * ==================================================================================== */

/* lincomb — mul.rs:112-163 (constant-time, GLV + radix-16, one shared doubling chain) */
static k256_pt k256_lincomb(const k256_pt *xs, const sc4 *ks, size_t n) {
    k256_lut *tables = (k256_lut *)ecref_xmalloc(sizeof(k256_lut) * 2 * (n ? n : 1));
    int8_t(*digits)[33] = (int8_t(*)[33])ecref_xmalloc(33 * 2 * (n ? n : 1));

    for (size_t i = 0; i < n; i++) {
        sc4 r1, r2;
        k256_decompose_scalar(&ks[i], &r1, &r2);
        k256_pt x_beta = k256_endomorphism(&xs[i]);
        int s1 = sc4_is_high(&r1), s2 = sc4_is_high(&r2);
        sc4 r1c = s1 ? sc4_neg(&r1) : r1;
        sc4 r2c = s2 ? sc4_neg(&r2) : r2;
        k256_pt p1 = s1 ? k256_neg(&xs[i]) : xs[i];
        k256_pt p2 = s2 ? k256_neg(&x_beta) : x_beta;
        k256_lut_new(&tables[2 * i], &p1);
        k256_lut_new(&tables[2 * i + 1], &p2);
        uint8_t be[32];
        sc4_to_be(&r1c, be); ecref_radix16(be, 32, 33, digits[2 * i]);
        sc4_to_be(&r2c, be); ecref_radix16(be, 32, 33, digits[2 * i + 1]);
    }

    k256_pt acc = k256_identity(), t;
    for (size_t c = 0; c < n; c++) {
        t = k256_lut_select(&tables[2 * c], digits[2 * c][32]);      acc = k256_add(&acc, &t);
        t = k256_lut_select(&tables[2 * c + 1], digits[2 * c + 1][32]); acc = k256_add(&acc, &t);
    }
    for (int i = 31; i >= 0; i--) {
        for (int j = 0; j < 4; j++) acc = k256_double(&acc);
        for (size_t c = 0; c < n; c++) {
            t = k256_lut_select(&tables[2 * c], digits[2 * c][i]);      acc = k256_add(&acc, &t);
            t = k256_lut_select(&tables[2 * c + 1], digits[2 * c + 1][i]); acc = k256_add(&acc, &t);
        }
    }
    free(tables);
    free(digits);
    return acc;
}

/* BASEPOINT_TABLE — tables.rs:11-18 + primeorder tables/basepoint.rs:41-76: 33 LUTs of
 * 2^(8i)*G */
static k256_lut K256_BASE_TABLE[33];
static int k256_base_table_ready;

static void k256_base_table_init(void) {
    if (k256_base_table_ready) return;
    k256_pt g = k256_generator();
    for (int i = 0; i < 33; i++) {
        k256_lut_new(&K256_BASE_TABLE[i], &g);
        for (int j = 0; j < 8; j++) g = k256_double(&g);
    }
    k256_base_table_ready = 1;
}

/* mul_by_generator — mul.rs:180-197 */
static k256_pt k256_mul_by_generator(const sc4 *k) {
    int8_t d[65];
    uint8_t be[32];
    sc4_to_be(k, be);
    ecref_radix16(be, 32, 65, d);
    k256_pt acc = k256_lut_select(&K256_BASE_TABLE[32], d[64]);
    k256_pt acc2 = k256_identity(), t;
    for (int i = 31; i >= 0; i--) {
        t = k256_lut_select(&K256_BASE_TABLE[i], d[2 * i + 1]); acc2 = k256_add(&acc2, &t);
        t = k256_lut_select(&K256_BASE_TABLE[i], d[2 * i]);     acc = k256_add(&acc, &t);
    }
    for (int j = 0; j < 4; j++) acc2 = k256_double(&acc2);
    return k256_add(&acc, &acc2);
}

/* ---- wNAF instantiation (wnaf crate, generic over the group) ---- */
#define WN_PT k256_pt
#define WN_PFX k256
#define WN_IDENTITY() k256_identity()
#define WN_DOUBLE(p) k256_double(p)
#define WN_ADD(a, b) k256_add(a, b)
#define WN_NEG(a) k256_neg(a)
#include "ecref_wnaf.inc"

/* glv::decompose_wnaf_into — mul/glv.rs:170-189; returns two (table, digits, len) terms */
static void k256_decompose_wnaf(const k256_pt *x, const sc4 *k, k256_wnaf_term out[2]) {
    sc4 r1, r2;
    k256_decompose_scalar(k, &r1, &r2);
    int n1 = sc4_is_high(&r1), n2 = sc4_is_high(&r2);
    if (n1) r1 = sc4_neg(&r1);
    if (n2) r2 = sc4_neg(&r2);
    uint8_t le[32];
    ecref_words_to_le(r1.w, 4, le);
    out[0].len = (size_t)ecref_wnaf_form(le, 16, 128, 5, out[0].digits);   /* GLV_LE_BYTES = 16 */
    ecref_words_to_le(r2.w, 4, le);
    out[1].len = (size_t)ecref_wnaf_form(le, 16, 128, 5, out[1].digits);
    k256_pt p1 = n1 ? k256_neg(x) : *x;
    k256_pt pb = k256_endomorphism(x);
    k256_pt p2 = n2 ? k256_neg(&pb) : pb;
    k256_wnaf_table(out[0].table, &p1);
    k256_wnaf_table(out[1].table, &p2);
}

/* lincomb_vartime — mul.rs:100-108,167-175 */
static k256_pt k256_lincomb_vartime(const k256_pt *xs, const sc4 *ks, size_t n) {
    k256_wnaf_term *terms = (k256_wnaf_term *)ecref_xmalloc(sizeof(k256_wnaf_term) * 2 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) k256_decompose_wnaf(&xs[i], &ks[i], &terms[2 * i]);
    k256_pt r = k256_wnaf_multi_exp(terms, 2 * n);
    free(terms);
    return r;
}

/* ======================================================================================
 * ABI glue (dispatch targets used by ecref.c)
 * ==================================================================================== */

static int k256_load_scalar(sc4 *k, const uint8_t *be) {
    *k = sc4_from_be(be);
    return ecref_mp_cmp(k->w, K256_N, 4) < 0;     /* from_repr rejects >= n, scalar.rs:310-316 */
}

int ecref_k256_batch_mul_base(const uint8_t *scalars, size_t n, uint8_t *out_xy, uint8_t *out_inf) {
    k256_base_table_init();
    for (size_t i = 0; i < n; i++) {
        sc4 k;
        if (!k256_load_scalar(&k, scalars + 32 * i)) return ECREF_ERR_SCALAR_RANGE;
        k256_pt r = k256_mul_by_generator(&k);
        k256_aff a = k256_to_affine(&r);
        k256_aff_to_bytes(&a, out_xy + 64 * i, out_inf ? out_inf + i : NULL);
    }
    return ECREF_OK;
}

int ecref_k256_batch_mul(const uint8_t *scalars, const uint8_t *pxy, const uint8_t *pinf, size_t n,
                         int vartime, uint8_t *out_xy, uint8_t *out_inf) {
    for (size_t i = 0; i < n; i++) {
        sc4 k;
        k256_aff a;
        if (!k256_load_scalar(&k, scalars + 32 * i)) return ECREF_ERR_SCALAR_RANGE;
        if (!k256_aff_from_bytes(&a, pxy + 64 * i, pinf ? pinf[i] : 0)) return ECREF_ERR_POINT;
        k256_pt p = k256_from_affine(&a);
        /* mul (mul.rs:236-238) = lincomb of one term; mul_vartime (mul.rs:242-247) */
        k256_pt r = vartime ? k256_lincomb_vartime(&p, &k, 1) : k256_lincomb(&p, &k, 1);
        k256_aff o = k256_to_affine(&r);
        k256_aff_to_bytes(&o, out_xy + 64 * i, out_inf ? out_inf + i : NULL);
    }
    return ECREF_OK;
}

int ecref_k256_msm(const uint8_t *scalars, const uint8_t *pxy, const uint8_t *pinf, size_t n,
                   size_t chunk, int vartime, uint8_t *out_xy, uint8_t *out_inf) {
    if (chunk == 0) chunk = 4096;
    k256_pt total = k256_identity();
    k256_pt *pts = (k256_pt *)ecref_xmalloc(sizeof(k256_pt) * chunk);
    sc4 *ks = (sc4 *)ecref_xmalloc(sizeof(sc4) * chunk);
    int rc = ECREF_OK;
    for (size_t off = 0; off < n && rc == ECREF_OK; off += chunk) {
        size_t m = n - off < chunk ? n - off : chunk;
        for (size_t i = 0; i < m; i++) {
            k256_aff a;
            if (!k256_load_scalar(&ks[i], scalars + 32 * (off + i))) { rc = ECREF_ERR_SCALAR_RANGE; break; }
            if (!k256_aff_from_bytes(&a, pxy + 64 * (off + i), pinf ? pinf[off + i] : 0)) { rc = ECREF_ERR_POINT; break; }
            pts[i] = k256_from_affine(&a);
        }
        if (rc != ECREF_OK) break;
        k256_pt part = vartime ? k256_lincomb_vartime(pts, ks, m) : k256_lincomb(pts, ks, m);
        total = k256_add(&total, &part);
    }
    free(pts);
    free(ks);
    if (rc != ECREF_OK) return rc;
    k256_aff o = k256_to_affine(&total);
    k256_aff_to_bytes(&o, out_xy, out_inf);
    return ECREF_OK;
}

int ecref_k256_mul_base_and_mul_add_vartime(const uint8_t *a, const uint8_t *b, const uint8_t *p_xy,
                                            int p_inf, uint8_t *out_xy, uint8_t *out_inf) {
    /* mul.rs:303-310: lincomb_vartime_glv_wnaf over [(G, a), (P, b)] */
    sc4 ks[2];
    k256_pt xs[2];
    k256_aff pa;
    if (!k256_load_scalar(&ks[0], a) || !k256_load_scalar(&ks[1], b)) return ECREF_ERR_SCALAR_RANGE;
    if (!k256_aff_from_bytes(&pa, p_xy, p_inf)) return ECREF_ERR_POINT;
    xs[0] = k256_generator();
    xs[1] = k256_from_affine(&pa);
    k256_pt r = k256_lincomb_vartime(xs, ks, 2);
    k256_aff o = k256_to_affine(&r);
    k256_aff_to_bytes(&o, out_xy, out_inf);
    return ECREF_OK;
}

/* DecompressPoint::decompress — affine.rs:261-280: alpha = x^3 + 7, beta = sqrt(alpha) normalized, y = beta or -beta
 * by parity.  ok[i] = 0 (and zeros out) when x >= p or alpha is a non-residue. */
int ecref_k256_batch_decompress(const uint8_t *xs, const uint8_t *y_is_odd, size_t n, uint8_t *out_xy, uint8_t *ok) {
    for (size_t i = 0; i < n; i++) {
        fe5 x, beta;
        ok[i] = 0;
        memset(out_xy + 64 * i, 0, 64);
        if (!fe5_from_bytes(&x, xs + 32 * i)) continue;
        fe5 xx = fe5_mul(&x, &x);
        fe5 x3 = fe5_mul(&xx, &x);
        fe5 b = FE5_ZERO;
        b.n[0] = K256_B_SINGLE;
        fe5 alpha = fe5_add(&x3, &b);
        if (!fe5_sqrt(&beta, &alpha)) continue;
        beta = fe5_normalize(&beta);
        int odd = (int)(beta.n[0] & 1);
        fe5 y = beta;
        if (odd != (y_is_odd[i] & 1)) { y = fe5_negate(&beta, 1); y = fe5_normalize(&y); }
        fe5_to_bytes(out_xy + 64 * i, &x);
        fe5_to_bytes(out_xy + 64 * i + 32, &y);
        ok[i] = 1;
    }
    return ECREF_OK;
}

int ecref_k256_field_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    fe5 x, y = FE5_ZERO, r;
    if (!fe5_from_bytes(&x, a)) return ECREF_ERR_POINT;
    if (b && !fe5_from_bytes(&y, b)) return ECREF_ERR_POINT;
    switch (op) {
    case 0: r = fe5_add(&x, &y); break;
    case 1: { fe5 ny = fe5_negate(&y, 1); r = fe5_add(&x, &ny); break; }
    case 2: r = fe5_mul(&x, &y); break;
    case 3: r = fe5_sqr(&x); break;
    case 4: if (!fe5_invert(&r, &x)) r = FE5_ZERO; break;
    case 5: r = fe5_negate(&x, 1); break;
    default: return ECREF_ERR_CURVE;
    }
    fe5_to_bytes(out, &r);
    return ECREF_OK;
}

int ecref_k256_point_op(int op, const uint8_t *p_xy, int p_inf, const uint8_t *q_xy, int q_inf,
                        uint8_t *out_xy, uint8_t *out_inf) {
    k256_aff pa, qa;
    if (!k256_aff_from_bytes(&pa, p_xy, p_inf)) return ECREF_ERR_POINT;
    k256_pt p = k256_from_affine(&pa), r;
    if (op == 0 || op == 1) {
        if (!k256_aff_from_bytes(&qa, q_xy, q_inf)) return ECREF_ERR_POINT;
        if (op == 0) { k256_pt q = k256_from_affine(&qa); r = k256_add(&p, &q); }
        else r = k256_add_mixed(&p, &qa);
    } else if (op == 2) r = k256_double(&p);
    else if (op == 3) r = k256_neg(&p);
    else return ECREF_ERR_CURVE;
    k256_aff o = k256_to_affine(&r);
    k256_aff_to_bytes(&o, out_xy, out_inf);
    return ECREF_OK;
}

int ecref_k256_batch_normalize(const uint8_t *xyz, size_t n, uint8_t *out_xy, uint8_t *out_inf) {
    k256_pt *pts = (k256_pt *)ecref_xmalloc(sizeof(k256_pt) * (n ? n : 1));
    k256_aff *aff = (k256_aff *)ecref_xmalloc(sizeof(k256_aff) * (n ? n : 1));
    fe5 *scratch = (fe5 *)ecref_xmalloc(sizeof(fe5) * (n ? n : 1));
    int rc = ECREF_OK;
    for (size_t i = 0; i < n && rc == ECREF_OK; i++) {
        if (!fe5_from_bytes(&pts[i].x, xyz + 96 * i) || !fe5_from_bytes(&pts[i].y, xyz + 96 * i + 32) ||
            !fe5_from_bytes(&pts[i].z, xyz + 96 * i + 64))
            rc = ECREF_ERR_POINT;
    }
    if (rc == ECREF_OK) {
        k256_batch_normalize(pts, n, aff, scratch);
        for (size_t i = 0; i < n; i++) k256_aff_to_bytes(&aff[i], out_xy + 64 * i, out_inf ? out_inf + i : NULL);
    }
    free(pts); free(aff); free(scratch);
    return rc;
}

int ecref_k256_validate_points(const uint8_t *pxy, const uint8_t *pinf, size_t n, size_t *bad) {
    for (size_t i = 0; i < n; i++) {
        k256_aff a;
        if (!k256_aff_from_bytes(&a, pxy + 64 * i, pinf ? pinf[i] : 0)) {
            if (bad) *bad = i;
            return ECREF_ERR_POINT;
        }
    }
    return ECREF_OK;
}

void ecref_k256_init(void) { k256_base_table_init(); }

void ecref_k256_scalar_reduce(uint8_t *scalars, size_t n) {
    for (size_t i = 0; i < n; i++) {
        sc4 k = sc4_from_be(scalars + 32 * i), t;
        if (!ecref_mp_sub(t.w, k.w, K256_N, 4)) sc4_to_be(&t, scalars + 32 * i);
    }
}
