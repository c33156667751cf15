/*
 * ecref.c — CPU ORACLE (test infrastructure, see ecref.h): curve dispatch plus the two
 * group-independent recodings.
 *
 *   Radix16Decomposition::new   primeorder/src/tables/radix16.rs:35-61
 *   wnaf_form / LimbBuffer      wnaf/src/lib.rs:70-150, wnaf/src/limb_buffer.rs:5-69
 */
#include "ecref_internal.h"

__attribute__((constructor)) static void ecref_init_all(void) {
    ecref_k256_init();
    ecref_p256_init();
    ecref_p384_init();
    ecref_sm2_init();
    ecref_p224_init();
    ecref_p192_init();
    ecref_p521_init();
    ecref_bp256_init();
    ecref_bp384_init();
    ecref_bp256t1_init();
    ecref_bp384t1_init();
    ecref_bign256_init();
}

size_t ecref_field_bytes(int curve) {
    switch (curve) {
    case ECREF_K256: case ECREF_P256: case ECREF_SM2: case ECREF_BP256: case ECREF_BP256T1: case ECREF_BIGN256: return 32;
    case ECREF_P224: return 28;
    case ECREF_P192: return 24;
    case ECREF_P521: return 66;
    case ECREF_BP384: case ECREF_BP384T1: return 48;
    case ECREF_P384: return 48;
    default: return 0;
    }
}

/* Radix16Decomposition::<D>::new — radix16.rs:35-61.
 * Step 1: nibbles of the low (D-1)/2 bytes of the big-endian repr, least significant first.
 * Step 2: recentre from [0,16) to [-8,8) with a carry into the next digit. */
int ecref_radix16(const uint8_t *scalar_be, size_t scalar_len, int ndigits, int8_t *digits) {
    memset(digits, 0, (size_t)ndigits);
    for (int i = 0; i < (ndigits - 1) / 2; i++) {
        uint8_t b = scalar_be[scalar_len - 1 - (size_t)i];
        digits[2 * i] = (int8_t)(b & 0xf);
        digits[2 * i + 1] = (int8_t)((b >> 4) & 0xf);
    }
    for (int i = 0; i < ndigits - 1; i++) {
        int8_t carry = (int8_t)((digits[i] + 8) >> 4);
        digits[i] = (int8_t)(digits[i] - (carry << 4));
        digits[i + 1] = (int8_t)(digits[i + 1] + carry);
    }
    return ECREF_OK;
}

/* LimbBuffer::get — limb_buffer.rs: u64 limb `idx` of the little-endian byte string, bytes
 * past the end read as zero. */
static uint64_t le_limb(const uint8_t *buf, size_t nbytes, size_t idx) {
    uint64_t v = 0;
    for (size_t j = 0; j < 8; j++) {
        size_t k = 8 * idx + j;
        if (k < nbytes) v |= (uint64_t)buf[k] << (8 * j);
    }
    return v;
}

/* wnaf_form — lib.rs:70-150 */
int ecref_wnaf_form(const uint8_t *le_bytes, size_t nbytes, size_t bit_len, int window,
                    int8_t *wnaf) {
    const uint64_t width = 1ULL << window;
    const uint64_t window_mask = width - 1;
    size_t pos = 0, cursor = 0;
    uint64_t carry = 0;

    while (pos < bit_len) {
        size_t u64_idx = pos / 64, bit_idx = pos % 64;
        uint64_t cur = le_limb(le_bytes, nbytes, u64_idx);
        uint64_t next = le_limb(le_bytes, nbytes, u64_idx + 1);
        uint64_t bit_buf;
        if (bit_idx + (size_t)window < 64) bit_buf = cur >> bit_idx;
        else bit_buf = (cur >> bit_idx) | (next << (64 - bit_idx));

        uint64_t window_val = carry + (bit_buf & window_mask);
        if ((window_val & 1) == 0) {
            wnaf[cursor++] = 0;
            pos += 1;
        } else {
            int8_t d = (int8_t)window_val;
            if (window_val < width / 2) {
                carry = 0;
            } else {
                carry = 1;
                d = (int8_t)(d - (int8_t)width);
            }
            wnaf[cursor++] = d;
            size_t max_pos = bit_len >= carry ? bit_len - (size_t)carry : 0;   /* saturating_sub */
            size_t skip = (size_t)window < max_pos - pos ? (size_t)window : max_pos - pos;
            for (size_t s = 1; s < skip; s++) wnaf[cursor++] = 0;
            pos += skip;
        }
    }
    if (carry != 0) wnaf[cursor++] = (int8_t)carry;
    return (int)cursor;
}

/* ---- little-endian wire format of bign-curve256v1 (bignp256/src/lib.rs:102, arithmetic/field.rs:65, arithmetic/scalar.rs:56) --
 * The curve sections all work on big-endian records; for ECREF_BIGN256 every record (scalar, coordinate) is byte-reversed on
 * the way in and on the way out.  The algorithms themselves read the scalar's integer value (to_be_repr / le_repr). */
#define BIGN_L 32
static uint8_t *bign_rev_dup(const uint8_t *src, size_t nrec) {
    if (!src) return NULL;
    uint8_t *d = (uint8_t *)ecref_xmalloc(nrec * BIGN_L + 1);
    for (size_t r = 0; r < nrec; r++)
        for (int j = 0; j < BIGN_L; j++) d[r * BIGN_L + j] = src[r * BIGN_L + BIGN_L - 1 - j];
    return d;
}
static void bign_rev_inplace(uint8_t *buf, size_t nrec) {
    if (!buf) return;
    for (size_t r = 0; r < nrec; r++)
        for (int j = 0; j < BIGN_L / 2; j++) {
            uint8_t t = buf[r * BIGN_L + j];
            buf[r * BIGN_L + j] = buf[r * BIGN_L + BIGN_L - 1 - j];
            buf[r * BIGN_L + BIGN_L - 1 - j] = t;
        }
}

#define DISPATCH(curve, fn, args)                     \
    switch (curve) {                                  \
    case ECREF_K256: return ecref_k256_##fn args;     \
    case ECREF_P256: return ecref_p256_##fn args;     \
    case ECREF_P384: return ecref_p384_##fn args;     \
    case ECREF_SM2: return ecref_sm2_##fn args;       \
    case ECREF_P224: return ecref_p224_##fn args;     \
    case ECREF_P192: return ecref_p192_##fn args;     \
    case ECREF_P521: return ecref_p521_##fn args;     \
    case ECREF_BP256: return ecref_bp256_##fn args;   \
    case ECREF_BP384: return ecref_bp384_##fn args;   \
    case ECREF_BP256T1: return ecref_bp256t1_##fn args;   \
    case ECREF_BP384T1: return ecref_bp384t1_##fn args;   \
    case ECREF_BIGN256: return ecref_bign256_##fn args;   \
    default: return ECREF_ERR_CURVE;                  \
    }

int ecref_batch_mul_base(int curve, const uint8_t *s, size_t n, uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *s2 = bign_rev_dup(s, n);
        int rc = ecref_bign256_batch_mul_base(s2, n, o, oi);
        bign_rev_inplace(o, 2 * n);
        free(s2);
        return rc;
    }
    DISPATCH(curve, batch_mul_base, (s, n, o, oi))
}
int ecref_batch_mul(int curve, const uint8_t *s, const uint8_t *p, const uint8_t *pi, size_t n,
                    uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *s2 = bign_rev_dup(s, n), *p2 = bign_rev_dup(p, 2 * n);
        int rc = ecref_bign256_batch_mul(s2, p2, pi, n, 0, o, oi);
        bign_rev_inplace(o, 2 * n);
        free(s2); free(p2);
        return rc;
    }
    DISPATCH(curve, batch_mul, (s, p, pi, n, 0, o, oi))
}
int ecref_batch_mul_vartime(int curve, const uint8_t *s, const uint8_t *p, const uint8_t *pi,
                            size_t n, uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *s2 = bign_rev_dup(s, n), *p2 = bign_rev_dup(p, 2 * n);
        int rc = ecref_bign256_batch_mul(s2, p2, pi, n, 1, o, oi);
        bign_rev_inplace(o, 2 * n);
        free(s2); free(p2);
        return rc;
    }
    DISPATCH(curve, batch_mul, (s, p, pi, n, 1, o, oi))
}
int ecref_msm(int curve, const uint8_t *s, const uint8_t *p, const uint8_t *pi, size_t n,
              size_t chunk, int vartime, uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *s2 = bign_rev_dup(s, n), *p2 = bign_rev_dup(p, 2 * n);
        int rc = ecref_bign256_msm(s2, p2, pi, n, chunk, vartime, o, oi);
        bign_rev_inplace(o, 2);
        free(s2); free(p2);
        return rc;
    }
    DISPATCH(curve, msm, (s, p, pi, n, chunk, vartime, o, oi))
}
int ecref_mul_base_and_mul_add_vartime(int curve, const uint8_t *a, const uint8_t *b,
                                       const uint8_t *p, int pi, uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *a2 = bign_rev_dup(a, 1), *b2 = bign_rev_dup(b, 1), *p2 = bign_rev_dup(p, 2);
        int rc = ecref_bign256_mul_base_and_mul_add_vartime(a2, b2, p2, pi, o, oi);
        bign_rev_inplace(o, 2);
        free(a2); free(b2); free(p2);
        return rc;
    }
    DISPATCH(curve, mul_base_and_mul_add_vartime, (a, b, p, pi, o, oi))
}
int ecref_batch_decompress(int curve, const uint8_t *xs, const uint8_t *odd, size_t n, uint8_t *o, uint8_t *ok) {
    if (curve == ECREF_BIGN256) {
        uint8_t *x2 = bign_rev_dup(xs, n);
        int rc = ecref_bign256_batch_decompress(x2, odd, n, o, ok);
        bign_rev_inplace(o, 2 * n);
        free(x2);
        return rc;
    }
    DISPATCH(curve, batch_decompress, (xs, odd, n, o, ok))
}
int ecref_field_op(int curve, int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    if (curve == ECREF_BIGN256) {
        uint8_t *a2 = bign_rev_dup(a, 1), *b2 = bign_rev_dup(b, 1);
        int rc = ecref_bign256_field_op(op, a2, b2, out);
        bign_rev_inplace(out, 1);
        free(a2); free(b2);
        return rc;
    }
    DISPATCH(curve, field_op, (op, a, b, out))
}
int ecref_point_op(int curve, int op, const uint8_t *p, int pi, const uint8_t *q, int qi,
                   uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *p2 = bign_rev_dup(p, 2), *q2 = bign_rev_dup(q, 2);
        int rc = ecref_bign256_point_op(op, p2, pi, q2, qi, o, oi);
        bign_rev_inplace(o, 2);
        free(p2); free(q2);
        return rc;
    }
    DISPATCH(curve, point_op, (op, p, pi, q, qi, o, oi))
}
int ecref_batch_normalize(int curve, const uint8_t *xyz, size_t n, uint8_t *o, uint8_t *oi) {
    if (curve == ECREF_BIGN256) {
        uint8_t *x2 = bign_rev_dup(xyz, 3 * n);
        int rc = ecref_bign256_batch_normalize(x2, n, o, oi);
        bign_rev_inplace(o, 2 * n);
        free(x2);
        return rc;
    }
    DISPATCH(curve, batch_normalize, (xyz, n, o, oi))
}
int ecref_validate_points(int curve, const uint8_t *p, const uint8_t *pi, size_t n, size_t *bad) {
    if (curve == ECREF_BIGN256) {
        uint8_t *p2 = bign_rev_dup(p, 2 * n);
        int rc = ecref_bign256_validate_points(p2, pi, n, bad);
        free(p2);
        return rc;
    }
    DISPATCH(curve, validate_points, (p, pi, n, bad))
}
int ecref_scalar_reduce(int curve, uint8_t *s, size_t n) {
    switch (curve) {
    case ECREF_K256: ecref_k256_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_P256: ecref_p256_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_P384: ecref_p384_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_SM2: ecref_sm2_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_P224: ecref_p224_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_P192: ecref_p192_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_P521: ecref_p521_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_BP256: ecref_bp256_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_BP384: ecref_bp384_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_BP256T1: ecref_bp256t1_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_BP384T1: ecref_bp384t1_scalar_reduce(s, n); return ECREF_OK;
    case ECREF_BIGN256: bign_rev_inplace(s, n); ecref_bign256_scalar_reduce(s, n); bign_rev_inplace(s, n); return ECREF_OK;
    default: return ECREF_ERR_CURVE;
    }
}
