/* ecref_ecdsa.c — CPU restatement of ECDSA verification over the oracle's own a*G + b*P driver.
 * TEST INFRASTRUCTURE ONLY (see ecref.h).
 *
 * The verification equation is not in /root/reference: it lives in the un-vendored crate `ecdsa` 0.17.0
 * (Cargo.lock:428-429), `hazmat::verify_prehashed`, which the reference instantiates at p256/src/ecdsa.rs:69,
 * p384/src/ecdsa.rs and k256/src/ecdsa.rs:99-106 (NORMALIZE_S = true) and tests with its own vectors
 * ({p256,p384,k256}/src/test_vectors/ecdsa.rs via `new_verification_test!`, p256/src/ecdsa.rs:161-164).  Its published
 * algorithm (SEC1 v2 section 4.1.4, FIPS 186-5 section 6.4.2) is restated here:
 *     z = digest as an integer, reduced mod n  (Reduce<FieldBytes>: one conditional subtraction,
 *         k256/src/arithmetic/scalar.rs:618-631)
 *     reject unless 1 <= r, s < n (Signature::from_scalars) and, for NORMALIZE_S curves, s <= (n-1)/2
 *     u1 = z/s, u2 = r/s mod n;  R = u1*G + u2*Q  (mul_by_generator_and_mul_add_vartime,
 *         primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310 -> ecref_mul_base_and_mul_add_vartime)
 *     accept iff R is not the identity and x(R) mod n == r.
 * Parity is pinned by the reference's 31 ECDSA vectors in tests/golden/ (accept) and by an independent big-integer
 * model (tests/pyec.py) on random accept / reject cases.
 *
 * Scalar arithmetic mod n: Montgomery multiplication on 64-bit words, inversion as a^(n-2).  (The reference inverts
 * with safegcd; the value is the same.) */
#include <stdlib.h>
#include <string.h>

#include "ecref.h"
#include "ecref_internal.h"

typedef unsigned __int128 u128;

/* group orders, little-endian 64-bit words: k256/src/lib.rs:71, p256/src/lib.rs:60, p384/src/lib.rs:73 */
static const uint64_t ORDER_K256[6] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull,
                                       0xFFFFFFFFFFFFFFFFull, 0, 0};
static const uint64_t ORDER_P256[6] = {0xF3B9CAC2FC632551ull, 0xBCE6FAADA7179E84ull, 0xFFFFFFFFFFFFFFFFull,
                                       0xFFFFFFFF00000000ull, 0, 0};
static const uint64_t ORDER_P384[6] = {0xECEC196ACCC52973ull, 0x581A0DB248B0A77Aull, 0xC7634D81F4372DDFull,
                                       0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};

static const uint64_t ORDER_P224[4] = {0x13DD29455C5C2A3Dull, 0xFFFF16A2E0B8F03Eull, 0xFFFFFFFFFFFFFFFFull,
                                       0x00000000FFFFFFFFull};       /* p224/src/lib.rs:50-55 */

static const uint64_t ORDER_P521[9] = {0xBB6FB71E91386409ull, 0x3BB5C9B8899C47AEull, 0x7FCC0148F709A5D0ull, 0x51868783BF2F966Bull, 0xFFFFFFFFFFFFFFFAull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0x00000000000001FFull};   /* p521/src/lib.rs:51-60 */
static const uint64_t ORDER_BP256[4] = {0x901E0E82974856A7ull, 0x8C397AA3B561A6F7ull, 0x3E660A909D838D71ull, 0xA9FB57DBA1EEA9BCull};   /* bp256/src/lib.rs:70 */
static const uint64_t ORDER_BP384[6] = {0x3B883202E9046565ull, 0xCF3AB6AF6B7FC310ull, 0x1F166E6CAC0425A7ull, 0x152F7109ED5456B3ull, 0x0F5D6F7E50E641DFull, 0x8CB91E82A3386D28ull};   /* bp384/src/lib.rs:73 */
static const uint64_t ORDER_SM2[4] = {0x53BBF40939D54123ull, 0x7203DF6B21C6052Bull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull};   /* sm2/src/lib.rs:86 */
static const uint64_t ORDER_P192[3] = {0x146BC9B1B4D22831ull, 0xFFFFFFFF99DEF836ull, 0xFFFFFFFFFFFFFFFFull};   /* p192/src/lib.rs:41 */

typedef struct {
    int nl;               /* 64-bit words */
    const uint64_t *n;
    uint64_t ninv;        /* -n^-1 mod 2^64 */
    uint64_t r2[9];       /* 2^(128 nl) mod n */
} modn_t;

static int geq(const uint64_t *a, const uint64_t *b, int nl) {
    for (int i = nl - 1; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return 1;
}
static void sub_n(uint64_t *a, const uint64_t *b, int nl) {
    uint64_t borrow = 0;
    for (int i = 0; i < nl; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
static int is_zero(const uint64_t *a, int nl) {
    uint64_t z = 0;
    for (int i = 0; i < nl; i++) z |= a[i];
    return z == 0;
}
/* a = 2a mod n */
static void dbl_mod(uint64_t *a, const modn_t *m) {
    uint64_t carry = 0;
    for (int i = 0; i < m->nl; i++) {
        uint64_t hi = a[i] >> 63;
        a[i] = (a[i] << 1) | carry;
        carry = hi;
    }
    if (carry || geq(a, m->n, m->nl)) sub_n(a, m->n, m->nl);
}
/* a = a + b mod n  (a, b < n) */
static void add_mod(uint64_t *a, const uint64_t *b, const modn_t *m) {
    uint64_t carry = 0;
    for (int i = 0; i < m->nl; i++) {
        u128 c = (u128)a[i] + b[i] + carry;
        a[i] = (uint64_t)c;
        carry = (uint64_t)(c >> 64);
    }
    if (carry || geq(a, m->n, m->nl)) sub_n(a, m->n, m->nl);
}
static void modn_init(modn_t *m, int curve) {
    m->nl = curve == ECREF_P384 || curve == ECREF_BP384 || curve == ECREF_BP384T1 ? 6 : curve == ECREF_P192 ? 3 : curve == ECREF_P521 ? 9 : 4;
    m->n = curve == ECREF_K256 ? ORDER_K256 : curve == ECREF_SM2 ? ORDER_SM2 : curve == ECREF_P256 ? ORDER_P256 : curve == ECREF_P224 ? ORDER_P224 : curve == ECREF_P192 ? ORDER_P192 : curve == ECREF_P521 ? ORDER_P521 : curve == ECREF_BP256 || curve == ECREF_BP256T1 ? ORDER_BP256 : curve == ECREF_BP384 || curve == ECREF_BP384T1 ? ORDER_BP384 : ORDER_P384;
    uint64_t x = m->n[0];                       /* Newton: x = n^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) x *= 2 - m->n[0] * x;
    m->ninv = 0 - x;
    memset(m->r2, 0, sizeof m->r2);
    m->r2[0] = 1;
    for (int i = 0; i < 128 * m->nl; i++) dbl_mod(m->r2, m);
}
/* r = a*b*2^(-64 nl) mod n */
static void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const modn_t *m) {
    uint64_t t[11] = {0};
    const int nl = m->nl;
    for (int i = 0; i < nl; i++) {
        u128 c = 0;
        for (int j = 0; j < nl; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[nl];
        t[nl] = (uint64_t)c;
        t[nl + 1] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * m->ninv;
        c = (u128)q * m->n[0] + t[0];
        c >>= 64;
        for (int j = 1; j < nl; j++) {
            c += (u128)q * m->n[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[nl];
        t[nl - 1] = (uint64_t)c;
        t[nl] = t[nl + 1] + (uint64_t)(c >> 64);
    }
    if (t[nl] || geq(t, m->n, nl)) sub_n(t, m->n, nl);
    memcpy(r, t, 8 * nl);
}
static void mul_mod(uint64_t *r, const uint64_t *a, const uint64_t *b, const modn_t *m) {
    uint64_t t[9];
    mont_mul(t, a, b, m);
    mont_mul(r, t, m->r2, m);
}
/* r = a^(n-2) mod n */
static void inv_mod(uint64_t *r, const uint64_t *a, const modn_t *m) {
    uint64_t e[9], acc[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0}, base[9];
    memcpy(e, m->n, 8 * m->nl);
    e[0] -= 2;                                   /* n is odd and > 2 */
    memcpy(base, a, 8 * m->nl);
    for (int i = 64 * m->nl - 1; i >= 0; i--) {
        mul_mod(acc, acc, acc, m);
        if ((e[i / 64] >> (i % 64)) & 1) mul_mod(acc, acc, base, m);
    }
    memcpy(r, acc, 8 * m->nl);
}
static void from_be(uint64_t *w, const uint8_t *b, int nl) { ecref_be_to_words_n(b, 8 * (size_t)nl, w, (size_t)nl); }
static void to_be(uint8_t *b, const uint64_t *w, int nl) { ecref_words_to_be_n(w, b, 8 * (size_t)nl); }
/* L-byte records that need not fill the words (P-224: 28 bytes in 4 words) */
static void from_be_len(uint64_t *w, const uint8_t *b, size_t len, int nl) { ecref_be_to_words_n(b, len, w, (size_t)nl); }
static void to_be_len(uint8_t *b, const uint64_t *w, size_t len) { ecref_words_to_be_n(w, b, len); }

int ecref_ecdsa_verify_batch(int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s, const uint8_t *q_xy,
                             size_t n, int reject_high_s, uint8_t *ok) {
    if (curve != ECREF_K256 && curve != ECREF_P256 && curve != ECREF_P384 && curve != ECREF_P224 && curve != ECREF_P192 && curve != ECREF_P521 && curve != ECREF_BP256 && curve != ECREF_BP384 && curve != ECREF_BP256T1 && curve != ECREF_BP384T1) return ECREF_ERR_CURVE;
    modn_t m;
    modn_init(&m, curve);
    const int nl = m.nl;
    const size_t L = curve == ECREF_P224 ? 28 : curve == ECREF_P521 ? 66 : 8 * (size_t)nl;
    for (size_t i = 0; i < n; i++) {
        uint64_t zw[9], rw[9], sw[9], w[9], u1[9], u2[9];
        ok[i] = 0;
        from_be_len(zw, z + L * i, L, nl);
        from_be_len(rw, r + L * i, L, nl);
        from_be_len(sw, s + L * i, L, nl);
        if (is_zero(rw, nl) || geq(rw, m.n, nl) || is_zero(sw, nl) || geq(sw, m.n, nl)) continue;
        if (reject_high_s) {
            uint64_t twice[9];
            memcpy(twice, sw, 8 * nl);
            uint64_t top = twice[nl - 1] >> 63;
            for (int k = nl - 1; k > 0; k--) twice[k] = (twice[k] << 1) | (twice[k - 1] >> 63);
            twice[0] <<= 1;
            if (top || geq(twice, m.n, nl)) continue;            /* 2s >= n  <=>  s > (n-1)/2 */
        }
        if (geq(zw, m.n, nl)) sub_n(zw, m.n, nl);
        inv_mod(w, sw, &m);
        mul_mod(u1, zw, w, &m);
        mul_mod(u2, rw, w, &m);
        uint8_t a[66], b[66], xy[132], inf = 0;
        to_be_len(a, u1, L);
        to_be_len(b, u2, L);
        if (ecref_mul_base_and_mul_add_vartime(curve, a, b, q_xy + 2 * L * i, 0, xy, &inf) != ECREF_OK) continue;
        if (inf) continue;
        uint64_t x[9];
        from_be_len(x, xy, L, nl);
        if (geq(x, m.n, nl)) sub_n(x, m.n, nl);                 /* x < p < 2n */
        ok[i] = memcmp(x, rw, 8 * nl) == 0;
    }
    return ECREF_OK;
}

/* Public-key recovery — `VerifyingKey::recover_from_prehash(prehash, &signature, recovery_id)` of the un-vendored crate
 * `ecdsa` 0.17.0 (Cargo.lock:428-429; recovery.rs), which the reference exercises with its own vectors at
 * k256/src/ecdsa.rs:170-262 (RECOVERY_TEST_VECTORS, the Ethereum example) and p256/tests/ecdsa.rs:20-25.  Its published
 * algorithm (SEC1 v2 section 4.1.6 for one (R, recovery id) candidate):
 *     (r, s) = the signature's scalars, both in [1, n-1];  z = bits2field(prehash) reduced mod n
 *     recovery id byte: bit 0 = y(R) is odd, bit 1 = x(R) was reduced (x(R) = r + n); values above 3 do not parse
 *     x = r, or r + n when bit 1 is set — `checked_add` on the curve's Uint, and the decompression below rejects x >= p
 *     R = AffinePoint::decompress(x, y_is_odd)                                (error if there is no such point)
 *     u1 = -(r^-1 z), u2 = r^-1 s;  pk = ProjectivePoint::lincomb(&[(G, u1), (R, u2)])
 *     vk = VerifyingKey::from_affine(pk)   (error for the identity);  vk.verify_prehash(prehash, signature)?  — which is
 *     where a curve with NORMALIZE_S (k256) rejects a high s.
 * ok[i] = 1 and out_xy[i] = the recovered key, or ok[i] = 0 and a zero record.  Not offered for sm2 and bign256
 * (not ECDSA curves). */
int ecref_ecdsa_recover_batch(int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s, const uint8_t *recid, size_t n,
                              int reject_high_s, uint8_t *out_xy, uint8_t *ok) {
    if (curve != ECREF_K256 && curve != ECREF_P256 && curve != ECREF_P384 && curve != ECREF_P224 && curve != ECREF_P192 && curve != ECREF_P521 && curve != ECREF_BP256 && curve != ECREF_BP384 && curve != ECREF_BP256T1 && curve != ECREF_BP384T1) return ECREF_ERR_CURVE;
    modn_t m;
    modn_init(&m, curve);
    const int nl = m.nl;
    const size_t L = curve == ECREF_P224 ? 28 : curve == ECREF_P521 ? 66 : 8 * (size_t)nl;
    for (size_t i = 0; i < n; i++) {
        uint64_t zw[9], rw[9], sw[9], xw[10], rinv[9], u1[9], u2[9];
        ok[i] = 0;
        memset(out_xy + 2 * L * i, 0, 2 * L);
        if (recid[i] > 3) continue;                                   /* RecoveryId::from_byte */
        from_be_len(zw, z + L * i, L, nl);
        from_be_len(rw, r + L * i, L, nl);
        from_be_len(sw, s + L * i, L, nl);
        if (is_zero(rw, nl) || geq(rw, m.n, nl) || is_zero(sw, nl) || geq(sw, m.n, nl)) continue;   /* Signature::from_scalars */
        if (geq(zw, m.n, nl)) sub_n(zw, m.n, nl);                     /* Reduce<FieldBytes>: one conditional subtraction (z < 2n) */
        /* x = r (+ n): one spare word for the carry; a value that does not fit the L wire bytes is >= p */
        memcpy(xw, rw, 8 * nl);
        xw[nl] = 0;
        if (recid[i] & 2) {
            uint64_t carry = 0;
            for (int k = 0; k < nl; k++) {
                u128 c = (u128)xw[k] + m.n[k] + carry;
                xw[k] = (uint64_t)c;
                carry = (uint64_t)(c >> 64);
            }
            xw[nl] = carry;
        }
        if (xw[nl]) continue;
        if (L < 8 * (size_t)nl && (xw[nl - 1] >> (8 * (L % 8)))) continue;       /* p521: above 66 bytes */
        uint8_t xb[66], odd = recid[i] & 1, rxy[132], dok = 0;
        to_be_len(xb, xw, L);
        if (ecref_batch_decompress(curve, xb, &odd, 1, rxy, &dok) != ECREF_OK || !dok) continue;
        inv_mod(rinv, rw, &m);
        mul_mod(u1, rinv, zw, &m);
        if (!is_zero(u1, nl)) {                                       /* u1 = -(r^-1 z) */
            uint64_t t[9];
            memcpy(t, m.n, 8 * nl);
            sub_n(t, u1, nl);
            memcpy(u1, t, 8 * nl);
        }
        mul_mod(u2, rinv, sw, &m);
        /* lincomb(&[(G, u1), (R, u2)]) through the oracle's LinearCombination restatement */
        uint8_t sc[2 * 66], pts[4 * 66], pk[132], inf = 0;
        to_be_len(sc, u1, L);
        to_be_len(sc + L, u2, L);
        {
            uint8_t one[66] = {0}, gi = 0;
            one[L - 1] = 1;
            if (ecref_batch_mul_base(curve, one, 1, pts, &gi) != ECREF_OK) continue;           /* the generator's affine bytes */
        }
        memcpy(pts + 2 * L, rxy, 2 * L);
        if (ecref_msm(curve, sc, pts, 0, 2, 0, 0, pk, &inf) != ECREF_OK) continue;
        if (inf) continue;                                            /* VerifyingKey::from_affine rejects the identity */
        uint8_t v = 0;
        if (ecref_ecdsa_verify_batch(curve, z + L * i, r + L * i, s + L * i, pk, 1, reject_high_s, &v) != ECREF_OK || !v) continue;
        memcpy(out_xy + 2 * L * i, pk, 2 * L);
        ok[i] = 1;
    }
    return ECREF_OK;
}

/* SM2DSA verification (GB/T 32918.2, draft-shen-sm2-ecdsa 5.3) on the prehash — `PrehashVerifier::verify_prehash`,
 * sm2/src/dsa/verifying.rs:138-171: e = the 32-byte digest SM3(ZA || M) reduced mod n (`Scalar::reduce`); r, s the signature
 * halves in [1, n-1] (`Signature` holds NonZeroScalars: sm2/src/dsa.rs); t = r + s mod n, reject t = 0;
 * (x1, y1) = s G + t Q (`ProjectivePoint::lincomb`); accept iff r == e + (x1 mod n) mod n.  Like the reference, the
 * identity is not rejected separately: `to_affine().x()` of the identity is 0. */
int ecref_sm2dsa_verify_batch(const uint8_t *e, const uint8_t *r, const uint8_t *s, const uint8_t *q_xy, size_t n, uint8_t *ok) {
    modn_t m;
    modn_init(&m, ECREF_SM2);
    for (size_t i = 0; i < n; i++) {
        uint64_t ew[4], rw[4], sw[4], t[4], x[4];
        ok[i] = 0;
        from_be(ew, e + 32 * i, 4);
        from_be(rw, r + 32 * i, 4);
        from_be(sw, s + 32 * i, 4);
        if (is_zero(rw, 4) || geq(rw, m.n, 4) || is_zero(sw, 4) || geq(sw, m.n, 4)) continue;
        if (geq(ew, m.n, 4)) sub_n(ew, m.n, 4);
        memcpy(t, rw, 32);
        add_mod(t, sw, &m);
        if (is_zero(t, 4)) continue;
        uint8_t a[32], b[32], xy[64], inf = 0;
        to_be(a, sw, 4);
        to_be(b, t, 4);
        if (ecref_mul_base_and_mul_add_vartime(ECREF_SM2, a, b, q_xy + 64 * i, 0, xy, &inf) != ECREF_OK) continue;
        if (inf) memset(xy, 0, 64);
        from_be(x, xy, 4);
        if (geq(x, m.n, 4)) sub_n(x, m.n, 4);                    /* x < p < 2n */
        add_mod(x, ew, &m);
        ok[i] = memcmp(x, rw, 32) == 0;
    }
    return ECREF_OK;
}

/* ---- SM3 (GB/T 32905-2016; the reference uses the un-vendored crate sm3 0.5.0, Cargo.lock:1330-1331) and SM2DSA verification
 * of a MESSAGE: `VerifyingKey::new(distid, public_key)` -> `hash_z` (sm2/src/distid.rs:21-44):
 *     Z = SM3(ENTL || ID || a || b || xG || yG || xA || yA), ENTL = bit length of ID as 2 big-endian bytes
 * `Verifier::verify(msg, sig)` -> `hash_msg` (sm2/src/dsa/verifying.rs:126-130): e = SM3(Z || M), then `verify_prehash`
 * (:138-171 = ecref_sm2dsa_verify_batch above).  Pinned by the reference's message-level vector sm2/tests/sm2dsa.rs:16-35
 * and by OpenSSL's SM3 in the tests. --------------------------------------------------------------------------------- */
static uint32_t rol32(uint32_t x, int n) { n &= 31; return n ? (x << n) | (x >> (32 - n)) : x; }
static void sm3_block(uint32_t v[8], const uint8_t b[64]) {
    uint32_t w[68], w1[64];
    for (int j = 0; j < 16; j++) w[j] = ((uint32_t)b[4 * j] << 24) | ((uint32_t)b[4 * j + 1] << 16) | ((uint32_t)b[4 * j + 2] << 8) | b[4 * j + 3];
    for (int j = 16; j < 68; j++) {
        uint32_t x = w[j - 16] ^ w[j - 9] ^ rol32(w[j - 3], 15);
        w[j] = (x ^ rol32(x, 15) ^ rol32(x, 23)) ^ rol32(w[j - 13], 7) ^ w[j - 6];
    }
    for (int j = 0; j < 64; j++) w1[j] = w[j] ^ w[j + 4];
    uint32_t a = v[0], bb = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6], h = v[7];
    for (int j = 0; j < 64; j++) {
        uint32_t t = j < 16 ? 0x79cc4519u : 0x7a879d8au;
        uint32_t ss1 = rol32(rol32(a, 12) + e + rol32(t, j), 7);
        uint32_t ss2 = ss1 ^ rol32(a, 12);
        uint32_t ff = j < 16 ? (a ^ bb ^ c) : ((a & bb) | (a & c) | (bb & c));
        uint32_t gg = j < 16 ? (e ^ f ^ g) : ((e & f) | (~e & g));
        uint32_t tt1 = ff + d + ss2 + w1[j];
        uint32_t tt2 = gg + h + ss1 + w[j];
        d = c; c = rol32(bb, 9); bb = a; a = tt1;
        h = g; g = rol32(f, 19); f = e; e = tt2 ^ rol32(tt2, 9) ^ rol32(tt2, 17);
    }
    v[0] ^= a; v[1] ^= bb; v[2] ^= c; v[3] ^= d; v[4] ^= e; v[5] ^= f; v[6] ^= g; v[7] ^= h;
}
static void sm3(uint8_t out[32], const uint8_t *msg, size_t len) {
    uint32_t v[8] = {0x7380166fu, 0x4914b2b9u, 0x172442d7u, 0xda8a0600u, 0xa96f30bcu, 0x163138aau, 0xe38dee4du, 0xb0fb0e4eu};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sm3_block(v, msg + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sm3_block(v, tail);
    if (tl == 128) sm3_block(v, tail + 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(v[i] >> 24); out[4 * i + 1] = (uint8_t)(v[i] >> 16); out[4 * i + 2] = (uint8_t)(v[i] >> 8); out[4 * i + 3] = (uint8_t)v[i]; }
}
int ecref_sm3(const uint8_t *msg, size_t len, uint8_t *out32) { sm3(out32, msg, len); return ECREF_OK; }

/* curve constants as the 32-byte big-endian strings `to_bytes()` yields: a = p - 3 (sm2/src/arithmetic.rs:53-54), b (:57-59),
 * the generator (:67-74) */
static const uint8_t SM2_A_B_G[128] = {
    0xFF,0xFF,0xFF,0xFE,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0x00,0x00,0x00,0x00,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFF,0xFC,
    0x28,0xE9,0xFA,0x9E,0x9D,0x9F,0x5E,0x34,0x4D,0x5A,0x9E,0x4B,0xCF,0x65,0x09,0xA7,0xF3,0x97,0x89,0xF5,0x15,0xAB,0x8F,0x92,0xDD,0xBC,0xBD,0x41,0x4D,0x94,0x0E,0x93,
    0x32,0xC4,0xAE,0x2C,0x1F,0x19,0x81,0x19,0x5F,0x99,0x04,0x46,0x6A,0x39,0xC9,0x94,0x8F,0xE3,0x0B,0xBF,0xF2,0x66,0x0B,0xE1,0x71,0x5A,0x45,0x89,0x33,0x4C,0x74,0xC7,
    0xBC,0x37,0x36,0xA2,0xF4,0xF6,0x77,0x9C,0x59,0xBD,0xCE,0xE3,0x6B,0x69,0x21,0x53,0xD0,0xA9,0x87,0x7C,0xC6,0x2A,0x47,0x40,0x02,0xDF,0x32,0xE5,0x21,0x39,0xF0,0xA0};

/* ok[i] = `VerifyingKey::new(distid, Q_i)?.verify(msg_i, sig_i)`: one distinguishing identifier for the batch (at most 8191
 * bytes: ENTL is a u16 bit count, distid.rs:22-26), keys as affine x||y, messages of one length, signatures r || s. */
int ecref_sm2dsa_verify_msg_batch(const uint8_t *distid, size_t distid_len, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len,
                                  const uint8_t *sigs, size_t n, uint8_t *ok) {
    if (distid_len > 8191) return ECREF_ERR_CURVE;
    size_t zlen = 2 + distid_len + 128 + 64;
    uint8_t *zin = (uint8_t *)malloc(zlen), *ein = (uint8_t *)malloc(32 + msg_len);
    if (!zin || !ein) { free(zin); free(ein); return ECREF_ERR_CURVE; }
    zin[0] = (uint8_t)((distid_len * 8) >> 8);
    zin[1] = (uint8_t)(distid_len * 8);
    if (distid_len) memcpy(zin + 2, distid, distid_len);
    memcpy(zin + 2 + distid_len, SM2_A_B_G, 128);
    for (size_t i = 0; i < n; i++) {
        uint8_t e[32];
        memcpy(zin + 2 + distid_len + 128, q_xy + 64 * i, 64);
        sm3(ein, zin, zlen);                                     /* Z */
        if (msg_len) memcpy(ein + 32, msgs + msg_len * i, msg_len);
        sm3(e, ein, 32 + msg_len);
        ecref_sm2dsa_verify_batch(e, sigs + 64 * i, sigs + 64 * i + 32, q_xy + 64 * i, 1, ok + i);
    }
    free(zin);
    free(ein);
    return ECREF_OK;
}

/* ---- belt-hash (STB 34.101.31-2020 §7.8; the reference uses the un-vendored crates belt-hash / belt-block, bignp256/Cargo.toml)
 * and bign verification, bignp256/src/ecdsa/verifying.rs:100-169.  The standard's algorithm restated byte by byte; pinned by the
 * reference's own signature vector (bignp256/tests/ecdsa.rs:21-46): that signature only verifies if the hash is right. */
static const uint8_t BELT_H[256] = {
    0xB1,0x94,0xBA,0xC8,0x0A,0x08,0xF5,0x3B,0x36,0x6D,0x00,0x8E,0x58,0x4A,0x5D,0xE4,0x85,0x04,0xFA,0x9D,0x1B,0xB6,0xC7,0xAC,0x25,0x2E,0x72,0xC2,0x02,0xFD,0xCE,0x0D,
    0x5B,0xE3,0xD6,0x12,0x17,0xB9,0x61,0x81,0xFE,0x67,0x86,0xAD,0x71,0x6B,0x89,0x0B,0x5C,0xB0,0xC0,0xFF,0x33,0xC3,0x56,0xB8,0x35,0xC4,0x05,0xAE,0xD8,0xE0,0x7F,0x99,
    0xE1,0x2B,0xDC,0x1A,0xE2,0x82,0x57,0xEC,0x70,0x3F,0xCC,0xF0,0x95,0xEE,0x8D,0xF1,0xC1,0xAB,0x76,0x38,0x9F,0xE6,0x78,0xCA,0xF7,0xC6,0xF8,0x60,0xD5,0xBB,0x9C,0x4F,
    0xF3,0x3C,0x65,0x7B,0x63,0x7C,0x30,0x6A,0xDD,0x4E,0xA7,0x79,0x9E,0xB2,0x3D,0x31,0x3E,0x98,0xB5,0x6E,0x27,0xD3,0xBC,0xCF,0x59,0x1E,0x18,0x1F,0x4C,0x5A,0xB7,0x93,
    0xE9,0xDE,0xE7,0x2C,0x8F,0x0C,0x0F,0xA6,0x2D,0xDB,0x49,0xF4,0x6F,0x73,0x96,0x47,0x06,0x07,0x53,0x16,0xED,0x24,0x7A,0x37,0x39,0xCB,0xA3,0x83,0x03,0xA9,0x8B,0xF6,
    0x92,0xBD,0x9B,0x1C,0xE5,0xD1,0x41,0x01,0x54,0x45,0xFB,0xC9,0x5E,0x4D,0x0E,0xF2,0x68,0x20,0x80,0xAA,0x22,0x7D,0x64,0x2F,0x26,0x87,0xF9,0x34,0x90,0x40,0x55,0x11,
    0xBE,0x32,0x97,0x13,0x43,0xFC,0x9A,0x48,0xA0,0x2A,0x88,0x5F,0x19,0x4B,0x09,0xA1,0x7E,0xCD,0xA4,0xD0,0x15,0x44,0xAF,0x8C,0xA5,0x84,0x50,0xBF,0x66,0xD2,0xE8,0x8A,
    0xA2,0xD7,0x46,0x52,0x42,0xA8,0xDF,0xB3,0x69,0x74,0xC5,0x51,0xEB,0x23,0x29,0x21,0xD4,0xEF,0xD9,0xB4,0x3A,0x62,0x28,0x75,0x91,0x14,0x10,0xEA,0x77,0x6C,0xDA,0x1D};

static uint32_t belt_word(const uint8_t *b) { return (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24; }
static void belt_put(uint8_t *b, uint32_t w) { b[0] = (uint8_t)w; b[1] = (uint8_t)(w >> 8); b[2] = (uint8_t)(w >> 16); b[3] = (uint8_t)(w >> 24); }
static uint32_t belt_g(uint32_t u, int r) {
    uint32_t v = (uint32_t)BELT_H[u & 0xff] | (uint32_t)BELT_H[(u >> 8) & 0xff] << 8 | (uint32_t)BELT_H[(u >> 16) & 0xff] << 16 | (uint32_t)BELT_H[u >> 24] << 24;
    return (v << r) | (v >> (32 - r));
}
/* y = belt-block(x) under the 32-byte key theta (§7.1.3: eight rounds, round keys K_1 .. K_56 = theta_1 .. theta_8 repeated) */
static void belt_block(uint8_t y[16], const uint8_t x[16], const uint8_t theta[32]) {
    uint32_t K[57], a = belt_word(x), b = belt_word(x + 4), c = belt_word(x + 8), d = belt_word(x + 12), e, t;
    for (int j = 1; j <= 56; j++) K[j] = belt_word(theta + 4 * ((j - 1) % 8));
    for (uint32_t i = 1; i <= 8; i++) {
        b ^= belt_g(a + K[7 * i - 6], 5);
        c ^= belt_g(d + K[7 * i - 5], 21);
        a -= belt_g(b + K[7 * i - 4], 13);
        e = belt_g(b + c + K[7 * i - 3], 21) ^ i;
        b += e;
        c -= e;
        d += belt_g(c + K[7 * i - 2], 13);
        b ^= belt_g(a + K[7 * i - 1], 21);
        c ^= belt_g(d + K[7 * i], 5);
        t = a; a = b; b = t;
        t = c; c = d; d = t;
        t = b; b = c; c = t;
    }
    belt_put(y, b); belt_put(y + 4, d); belt_put(y + 8, a); belt_put(y + 12, c);
}
/* sigma1(u1 || u2 || u3 || u4) = belt-block(u3 ^ u4, u1 || u2) ^ u3 ^ u4 */
static void belt_sigma1(uint8_t out[16], const uint8_t u[64]) {
    uint8_t t[16];
    for (int j = 0; j < 16; j++) t[j] = u[32 + j] ^ u[48 + j];
    belt_block(out, t, u);
    for (int j = 0; j < 16; j++) out[j] ^= t[j];
}
/* sigma2(u) = (belt-block(u1, sigma1(u) || u4) ^ u1) || (belt-block(u2, (sigma1(u) ^ 1^128) || u3) ^ u2) */
static void belt_sigma2(uint8_t out[32], const uint8_t u[64]) {
    uint8_t s1[16], th[32];
    belt_sigma1(s1, u);
    memcpy(th, s1, 16);
    memcpy(th + 16, u + 48, 16);
    belt_block(out, u, th);
    for (int j = 0; j < 16; j++) { out[j] ^= u[j]; th[j] = (uint8_t)~s1[j]; }
    memcpy(th + 16, u + 32, 16);
    belt_block(out + 16, u + 16, th);
    for (int j = 0; j < 16; j++) out[16 + j] ^= u[16 + j];
}
void ecref_belt_hash(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint8_t u[64], s[16] = {0}, t[16], h[32];
    memcpy(h, BELT_H, 32);
    for (size_t off = 0; off < len; off += 32) {
        size_t m = len - off < 32 ? len - off : 32;
        memset(u, 0, 32);
        memcpy(u, msg + off, m);
        memcpy(u + 32, h, 32);
        belt_sigma1(t, u);
        for (int j = 0; j < 16; j++) s[j] ^= t[j];
        belt_sigma2(h, u);
    }
    memset(u, 0, 16);
    for (int j = 0; j < 8; j++) u[j] = (uint8_t)(((uint64_t)len * 8) >> (8 * j));
    memcpy(u + 16, s, 16);
    memcpy(u + 32, h, 32);
    belt_sigma2(out, u);
}

static const uint64_t ORDER_BIGN256[4] = {0x7E5ABF99263D6607ull, 0xD95C8ED60DFB4DFCull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};   /* bignp256/src/lib.rs:74 */
static const uint8_t BELT_OID[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};                                 /* bignp256/src/ecdsa.rs:58-60 */
static void from_le32(uint64_t w[4], const uint8_t *b, size_t len) {
    memset(w, 0, 32);
    for (size_t j = 0; j < len; j++) w[j / 8] |= (uint64_t)b[j] << (8 * (j % 8));
}
static void to_le32(uint8_t b[32], const uint64_t w[4]) {
    for (int j = 0; j < 32; j++) b[j] = (uint8_t)(w[j / 8] >> (8 * (j % 8)));
}
/* ok[i] = `VerifyingKey::from_bytes(Q_i)?.verify_prehash(h_i, &Signature::from_bytes(sig_i)?)`:
 *   Signature::from_bytes (bignp256/src/ecdsa.rs:72-88): S0 = the first 16 bytes, S1 = the other 32, little-endian; S1 >= q, S0 = 0 or
 *   S1 = 0 do not parse;  verifying.rs:100-147: R = ((S1 + H) mod q) G + (S0 + 2^128) Q, reject R = O,
 *   t = belt-hash(OID || x(R) || h), accept iff S0 == t[..16]. */
int ecref_bign_verify_batch(const uint8_t *h, const uint8_t *sigs, const uint8_t *q_xy, size_t n, uint8_t *ok) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t *sig = sigs + 48 * i;
        uint64_t s0[4], s1[4], hw[4];
        ok[i] = 0;
        from_le32(s0, sig, 16);
        from_le32(s1, sig + 16, 32);
        from_le32(hw, h + 32 * i, 32);
        if (is_zero(s0, 4) || is_zero(s1, 4) || geq(s1, ORDER_BIGN256, 4)) continue;
        if (geq(hw, ORDER_BIGN256, 4)) sub_n(hw, ORDER_BIGN256, 4);          /* Scalar::reduce: H < 2^256 < 2q */
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {                                         /* left = S1 + H mod q */
            u128 c = (u128)s1[j] + hw[j] + carry;
            s1[j] = (uint64_t)c;
            carry = (uint64_t)(c >> 64);
        }
        if (carry || geq(s1, ORDER_BIGN256, 4)) sub_n(s1, ORDER_BIGN256, 4);
        s0[2] += 1;                                                           /* right = S0 + 2^128 (< q) */
        uint8_t a[32], b[32], xy[64], inf = 0, msg[11 + 32 + 32], t[32];
        to_le32(a, s1);
        to_le32(b, s0);
        if (ecref_mul_base_and_mul_add_vartime(ECREF_BIGN256, a, b, q_xy + 64 * i, 0, xy, &inf) != ECREF_OK) continue;   /* key not on the curve */
        if (inf) continue;
        memcpy(msg, BELT_OID, 11);
        memcpy(msg + 11, xy, 32);
        memcpy(msg + 43, h + 32 * i, 32);
        ecref_belt_hash(msg, sizeof msg, t);
        ok[i] = memcmp(t, sig, 16) == 0;
    }
    return ECREF_OK;
}
/* ok[i] = `VerifyingKey::from_bytes(Q_i)?.verify(msg_i, &sig_i)` (verifying.rs:157-169): h = belt-hash(msg), then the above */
int ecref_bign_verify_msg_batch(const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n, uint8_t *ok) {
    for (size_t i = 0; i < n; i++) {
        uint8_t h[32];
        ecref_belt_hash(msgs + msg_len * i, msg_len, h);
        ecref_bign_verify_batch(h, sigs + 48 * i, q_xy + 64 * i, 1, ok + i);
    }
    return ECREF_OK;
}

/* BIP340 Schnorr verification over secp256k1 — `VerifyingKey::verify_raw`, k256/src/schnorr/verifying.rs:76-99, without
 * the hash: e is the challenge tagged_hash("BIP0340/challenge", r || pk || m) as 32 bytes (reduced mod n here like
 * `<Scalar as Reduce<FieldBytes>>::reduce`), (r, s) the signature halves parsed as in k256/src/schnorr.rs:132-150
 * (r < p, 0 < s < n), p_xy the verifying key's affine point.  R = s*G + (-e)*P
 * (mul_by_generator_and_mul_add_vartime); accept iff R is not the identity, y(R) is even and x(R) == r. */
static const uint64_t K256_FIELD_P[4] = {0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull,
                                         0xFFFFFFFFFFFFFFFFull};     /* k256/src/arithmetic/field.rs:41-42 */

int ecref_schnorr_verify_batch(const uint8_t *e, const uint8_t *r, const uint8_t *s, const uint8_t *p_xy, size_t n,
                               uint8_t *ok) {
    modn_t m;
    modn_init(&m, ECREF_K256);
    for (size_t i = 0; i < n; i++) {
        uint64_t ew[4], rw[4], sw[4], ne[4];
        ok[i] = 0;
        from_be(ew, e + 32 * i, 4);
        from_be(rw, r + 32 * i, 4);
        from_be(sw, s + 32 * i, 4);
        if (geq(rw, K256_FIELD_P, 4)) continue;
        if (is_zero(sw, 4) || geq(sw, m.n, 4)) continue;
        if (geq(ew, m.n, 4)) sub_n(ew, m.n, 4);
        memcpy(ne, m.n, 32);
        if (is_zero(ew, 4)) memset(ne, 0, 32); else sub_n(ne, ew, 4);
        uint8_t a[32], b[32], xy[64], inf = 0;
        to_be(a, sw, 4);
        to_be(b, ne, 4);
        if (ecref_mul_base_and_mul_add_vartime(ECREF_K256, a, b, p_xy + 64 * i, 0, xy, &inf) != ECREF_OK) continue;
        if (inf || (xy[63] & 1)) continue;
        ok[i] = memcmp(xy, r + 32 * i, 32) == 0;
    }
    return ECREF_OK;
}

/* ---- SHA-256 (FIPS 180-4) and the BIP340 tagged hash: k256/src/schnorr.rs `tagged_hash` = SHA256(SHA256(tag) ||
 * SHA256(tag) || ...), used by verify_raw (verifying.rs:79-85) with tag "BIP0340/challenge" ---------------------------- */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint8_t b[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++)
        w[i] = w[i - 16] + (ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
               (ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
        uint32_t t2 = (ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha256(uint8_t out[32], const uint8_t *msg, size_t len) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, msg + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

/* `VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)` for a batch of equally long messages — verifying.rs:76-99 (verify_raw),
 * :149-160 (from_bytes: lift_x through `AffinePoint::decompact` = decompress with even y), schnorr.rs:132-150 (signature
 * parsing).  pk_x n*32, msgs n*msg_len, sigs n*64 (r || s). */
int ecref_schnorr_verify_raw_batch(const uint8_t *pk_x, const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n,
                                   uint8_t *ok) {
    static const char TAG[] = "BIP0340/challenge";
    uint8_t th[32];
    sha256(th, (const uint8_t *)TAG, sizeof TAG - 1);
    uint8_t *buf = (uint8_t *)malloc(128 + msg_len);
    if (!buf) return ECREF_ERR_CURVE;
    for (size_t i = 0; i < n; i++) {
        uint8_t pxy[64], lifted, zero = 0, e[32];
        ok[i] = 0;
        if (ecref_batch_decompress(ECREF_K256, pk_x + 32 * i, &zero, 1, pxy, &lifted) != ECREF_OK || !lifted) continue;
        memcpy(buf, th, 32);
        memcpy(buf + 32, th, 32);
        memcpy(buf + 64, sigs + 64 * i, 32);
        memcpy(buf + 96, pk_x + 32 * i, 32);
        if (msg_len) memcpy(buf + 128, msgs + msg_len * i, msg_len);
        sha256(e, buf, 128 + msg_len);
        ecref_schnorr_verify_batch(e, sigs + 64 * i, sigs + 64 * i + 32, pxy, 1, ok + i);
    }
    free(buf);
    return ECREF_OK;
}

/* ---- SHA-224 / SHA-384 / SHA-512 (FIPS 180-4; the reference takes them from the un-vendored crate sha2) and ECDSA
 * verification of a MESSAGE: `signature::Verifier::verify(msg, &sig)` of `ecdsa::VerifyingKey<C>` hashes with the curve's
 * `DigestAlgorithm` (k256/src/ecdsa.rs:117-119 and p256/src/ecdsa.rs:72-74: Sha256; p384/src/ecdsa.rs:69-71: Sha384;
 * p224/src/ecdsa.rs:69-71: Sha224; p521/src/ecdsa.rs:69-71: Sha512; bp256 / bp384: Sha256 / Sha384), converts with
 * `bits2field` (leftmost L bytes, left-padded when shorter) and calls `verify_prehashed` (above).  Pinned by hashlib and by the
 * reference's Wycheproof blobs at the message level in the tests. ------------------------------------------------------- */
static const uint64_t SHA512_K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
        0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
        0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
        0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
        0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
        0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
        0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
        0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
        0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
        0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
        0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
        0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
        0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
        0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
        0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
        0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
        0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
static uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void sha512_block(uint64_t h[8], const uint8_t b[128]) {
    uint64_t w[80];
    for (int i = 0; i < 16; i++) {
        w[i] = 0;
        for (int k = 0; k < 8; k++) w[i] = (w[i] << 8) | b[8 * i + k];
    }
    for (int i = 16; i < 80; i++)
        w[i] = w[i - 16] + (ror64(w[i - 15], 1) ^ ror64(w[i - 15], 8) ^ (w[i - 15] >> 7)) + w[i - 7] +
               (ror64(w[i - 2], 19) ^ ror64(w[i - 2], 61) ^ (w[i - 2] >> 6));
    uint64_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; i++) {
        uint64_t t1 = hh + (ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41)) + ((e & f) ^ (~e & g)) + SHA512_K[i] + w[i];
        uint64_t t2 = (ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
/* digest_len 48 (SHA-384) or 64 (SHA-512) */
static void sha512_family(uint8_t *out, size_t digest_len, const uint8_t *msg, size_t len) {
    static const uint64_t IV512[8] = {
        0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint64_t IV384[8] = {
        0xcbbb9d5dc1059ed8ULL, 0x629a292a367cd507ULL, 0x9159015a3070dd17ULL, 0x152fecd8f70e5939ULL,
        0x67332667ffc00b31ULL, 0x8eb44a8768581511ULL, 0xdb0c2e0d64f98fa7ULL, 0x47b5481dbefa4fa4ULL};
    uint64_t h[8];
    memcpy(h, digest_len == 48 ? IV384 : IV512, sizeof h);
    size_t full = len / 128;
    for (size_t i = 0; i < full; i++) sha512_block(h, msg + 128 * i);
    uint8_t tail[256] = {0};
    size_t rem = len - 128 * full;
    memcpy(tail, msg + 128 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 17 <= 128 ? 128 : 256;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha512_block(h, tail);
    if (tl == 256) sha512_block(h, tail + 128);
    for (size_t i = 0; i < digest_len; i++) out[i] = (uint8_t)(h[i / 8] >> (8 * (7 - i % 8)));
}
/* SHA-224: SHA-256 with its own initial value, truncated to 28 bytes */
static void sha224(uint8_t out[28], const uint8_t *msg, size_t len) {
    uint32_t h[8] = {
        0xc1059ed8u, 0x367cd507u, 0x3070dd17u, 0xf70e5939u, 0xffc00b31u, 0x68581511u, 0x64f98fa7u, 0xbefa4fa4u};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, msg + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    memcpy(tail, msg + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 7; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}
/* the digest the reference binds to the curve; returns its length, 0 if the curve has none */
static size_t curve_digest(int curve, uint8_t *out, const uint8_t *msg, size_t len) {
    switch (curve) {
    case ECREF_K256: case ECREF_P256: case ECREF_BP256: case ECREF_BP256T1: sha256(out, msg, len); return 32;
    case ECREF_P384: case ECREF_BP384: case ECREF_BP384T1: sha512_family(out, 48, msg, len); return 48;
    case ECREF_P224: sha224(out, msg, len); return 28;
    case ECREF_P521: sha512_family(out, 64, msg, len); return 64;
    default: return 0;
    }
}
int ecref_curve_digest(int curve, const uint8_t *msg, size_t len, uint8_t *out, size_t *out_len) {
    *out_len = curve_digest(curve, out, msg, len);
    return *out_len ? ECREF_OK : ECREF_ERR_CURVE;
}
/* ok[i] = `VerifyingKey::from_affine(Q_i)?.verify(msg_i, sig_i)`: keys n*2L, messages of one length, sigs n*2L (r || s) */
int ecref_ecdsa_verify_msg_batch(int curve, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n,
                                 int reject_high_s, uint8_t *ok) {
    const size_t L = ecref_field_bytes(curve);
    uint8_t digest[64], z[66];
    if (!L || !curve_digest(curve, digest, (const uint8_t *)"", 0)) return ECREF_ERR_CURVE;
    for (size_t i = 0; i < n; i++) {
        size_t d = curve_digest(curve, digest, msgs + msg_len * i, msg_len);
        memset(z, 0, sizeof z);
        if (d >= L) memcpy(z, digest, L);                        /* bits2field: the leftmost L bytes ... */
        else memcpy(z + (L - d), digest, d);                     /* ... or left-padded */
        int rc = ecref_ecdsa_verify_batch(curve, z, sigs + 2 * L * i, sigs + 2 * L * i + L, q_xy + 2 * L * i, 1, reject_high_s, ok + i);
        if (rc != ECREF_OK) return rc;
    }
    return ECREF_OK;
}
