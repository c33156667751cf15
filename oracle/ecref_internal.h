/* ecref_internal.h — shared helpers of the CPU oracle (test infrastructure, see ecref.h). */
#ifndef ECREF_INTERNAL_H
#define ECREF_INTERNAL_H

#include "ecref.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

typedef struct { uint64_t n[5]; } fe5;     /* k256 FieldElement5x52 (field_5x52.rs:17) */

static inline void *ecref_xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "ecref: out of memory\n"); abort(); }
    return p;
}

/* big-endian bytes <-> little-endian 64-bit words (U256::from_be_slice / to_be_byte_array) */
static inline void ecref_be_to_words(const uint8_t *b, size_t len, uint64_t *w) {
    size_t nw = len / 8;
    for (size_t i = 0; i < nw; i++) {
        uint64_t v = 0;
        const uint8_t *p = b + len - 8 * (i + 1);
        for (int j = 0; j < 8; j++) v = (v << 8) | p[j];
        w[i] = v;
    }
}
static inline void ecref_words_to_be(const uint64_t *w, size_t nw, uint8_t *b) {
    for (size_t i = 0; i < nw; i++) {
        uint64_t v = w[i];
        uint8_t *p = b + 8 * (nw - 1 - i);
        for (int j = 7; j >= 0; j--) { p[j] = (uint8_t)v; v >>= 8; }
    }
}
/* the same for byte lengths that are not a multiple of 8 (P-224: 28 bytes in 4 words) */
static inline void ecref_be_to_words_n(const uint8_t *b, size_t len, uint64_t *w, size_t nw) {
    for (size_t i = 0; i < nw; i++) w[i] = 0;
    for (size_t i = 0; i < len && i < 8 * nw; i++) w[i / 8] |= (uint64_t)b[len - 1 - i] << (8 * (i % 8));
}
static inline void ecref_words_to_be_n(const uint64_t *w, uint8_t *b, size_t len) {
    for (size_t i = 0; i < len; i++) b[len - 1 - i] = (uint8_t)(w[i / 8] >> (8 * (i % 8)));
}
static inline void ecref_words_to_le(const uint64_t *w, size_t nw, uint8_t *b) {
    for (size_t i = 0; i < nw; i++)
        for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

/* multi-precision helpers on little-endian word arrays */
static inline uint64_t ecref_mp_add(uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n) {
    u128 c = 0;
    for (size_t i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t ecref_mp_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n) {
    uint64_t borrow = 0;
    for (size_t i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline int ecref_mp_cmp(const uint64_t *a, const uint64_t *b, size_t n) {
    for (size_t i = n; i-- > 0;) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return -1;
    }
    return 0;
}
static inline int ecref_mp_is_zero(const uint64_t *a, size_t n) {
    uint64_t z = 0;
    for (size_t i = 0; i < n; i++) z |= a[i];
    return z == 0;
}
/* r[0..na+nb) = a * b (schoolbook) */
static inline void ecref_mp_mul_rect(uint64_t *r, const uint64_t *a, size_t na, const uint64_t *b,
                                     size_t nb) {
    memset(r, 0, sizeof(uint64_t) * (na + nb));
    for (size_t i = 0; i < na; i++) {
        u128 c = 0;
        for (size_t j = 0; j < nb; j++) {
            c += (u128)a[i] * b[j] + r[i + j];
            r[i + j] = (uint64_t)c;
            c >>= 64;
        }
        r[i + nb] = (uint64_t)c;
    }
}
static inline void ecref_mp_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, size_t n) {
    ecref_mp_mul_rect(r, a, n, b, n);
}

/* per-curve entry points implemented in ecref_k256.c / ecref_prime.c, dispatched by ecref.c */
#define ECREF_DECL_CURVE(pfx)                                                                       \
    int ecref_##pfx##_batch_mul_base(const uint8_t *, size_t, uint8_t *, uint8_t *);                \
    int ecref_##pfx##_batch_mul(const uint8_t *, const uint8_t *, const uint8_t *, size_t, int,     \
                                uint8_t *, uint8_t *);                                              \
    int ecref_##pfx##_msm(const uint8_t *, const uint8_t *, const uint8_t *, size_t, size_t, int,   \
                          uint8_t *, uint8_t *);                                                    \
    int ecref_##pfx##_mul_base_and_mul_add_vartime(const uint8_t *, const uint8_t *,                \
                                                   const uint8_t *, int, uint8_t *, uint8_t *);     \
    int ecref_##pfx##_field_op(int, const uint8_t *, const uint8_t *, uint8_t *);                   \
    int ecref_##pfx##_batch_decompress(const uint8_t *, const uint8_t *, size_t, uint8_t *, uint8_t *); \
    int ecref_##pfx##_point_op(int, const uint8_t *, int, const uint8_t *, int, uint8_t *,          \
                               uint8_t *);                                                          \
    int ecref_##pfx##_batch_normalize(const uint8_t *, size_t, uint8_t *, uint8_t *);               \
    int ecref_##pfx##_validate_points(const uint8_t *, const uint8_t *, size_t, size_t *);          \
    void ecref_##pfx##_scalar_reduce(uint8_t *, size_t);                                            \
    void ecref_##pfx##_init(void);

ECREF_DECL_CURVE(k256)
ECREF_DECL_CURVE(p256)
ECREF_DECL_CURVE(p384)
ECREF_DECL_CURVE(sm2)
ECREF_DECL_CURVE(p224)
ECREF_DECL_CURVE(p192)
ECREF_DECL_CURVE(p521)
ECREF_DECL_CURVE(bp256)
ECREF_DECL_CURVE(bp384)
ECREF_DECL_CURVE(bp256t1)
ECREF_DECL_CURVE(bp384t1)
ECREF_DECL_CURVE(bign256)

#endif
