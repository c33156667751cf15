/*
 * ecref.h — CPU ORACLE for the ecgpu hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library (liboracle_ecref.so) is a plain-C restatement of the CPU algorithms of
 * RustCrypto/elliptic-curves for batch scalar multiplication / MSM on k256, p256 and p384.
 * It exists so that the HIP path can be checked bit-for-bit and so that bench.py has a
 * "reference algorithm on host cores" number to print beside the GPU number.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (libecgpu.so) never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this code against every golden
 * vector the reference holds for the path (k256/p256/p384 test_vectors/group.rs ADD+MUL
 * vectors, p256/p384/k256 test_vectors/ecdsa.rs d->Q and k->r vectors, the radix-16 unit-test
 * properties of primeorder/src/tables/radix16.rs:109-172) extracted into tests/golden/ by
 * tests/golden/extract_golden.py, and against an independent big-integer affine model.
 *
 * The reference itself (Rust, ~150 crates.io dependencies, no rustc in this image) cannot be
 * built here, so there is no oracle/_ref; arithmetic that lives in the un-vendored dependency
 * crypto-bigint 0.7.5 (Cargo.lock:367-368: widening_mul, ConstMontyForm mul/add/sub/neg/invert)
 * is restated from its published algorithm (word-by-word Montgomery multiplication, modular
 * inverse) and is pinned at the canonical-bytes boundary by the vectors above.
 *
 * Wire format (identical to include/ecgpu.h): scalars are n*L bytes big-endian (L = 32 for
 * k256/p256, 48 for p384), canonical (< group order n); points are n*2L bytes big-endian
 * affine x||y plus an optional n-byte infinity-flag array (NULL = no identities); the identity
 * is encoded x = y = 0, flag = 1 (k256/src/arithmetic/affine.rs:53-57,
 * primeorder/src/affine.rs:45-49).
 */
#ifndef ECREF_H
#define ECREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ECREF_K256 = 0, ECREF_P256 = 1, ECREF_P384 = 2, ECREF_SM2 = 3, ECREF_P224 = 4, ECREF_P192 = 5, ECREF_P521 = 6, ECREF_BP256 = 7, ECREF_BP384 = 8, ECREF_BP256T1 = 9, ECREF_BP384T1 = 10, ECREF_BIGN256 = 11 };

enum {
    ECREF_OK = 0,
    ECREF_ERR_CURVE = -1,
    ECREF_ERR_SCALAR_RANGE = -2,  /* scalar >= n   (k256 scalar.rs:310-316) */
    ECREF_ERR_POINT = -3          /* not on curve / coordinate >= p (primeorder affine.rs:100-109) */
};

/* Field-byte length L of the curve (32 or 48), or 0 for a bad id. */
size_t ecref_field_bytes(int curve);

/* ---- the three drivers of the path, constant-time reference algorithms ------------------ */

/* out[i] = mul_by_generator(k[i])  — basepoint-table path
 * (k256 mul.rs:180-197; primeorder tables/basepoint.rs:82-99). */
int ecref_batch_mul_base(int curve, const uint8_t *scalars, size_t n,
                         uint8_t *out_xy, uint8_t *out_inf);

/* out[i] = P[i] * k[i]  — constant-time LUT + radix-16 path
 * (k256 mul.rs:112-163 with N=1 incl. GLV; primeorder projective.rs:133-137,532-557). */
int ecref_batch_mul(int curve, const uint8_t *scalars, const uint8_t *points_xy,
                    const uint8_t *points_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);

/* out[i] = P[i].mul_vartime(k[i]) — wNAF-5 (+GLV on k256) path
 * (k256 mul.rs:242-247; primeorder projective.rs:142-144). */
int ecref_batch_mul_vartime(int curve, const uint8_t *scalars, const uint8_t *points_xy,
                            const uint8_t *points_inf, size_t n, uint8_t *out_xy,
                            uint8_t *out_inf);

/* out = sum_i k[i]*P[i] — LinearCombination::lincomb (vartime=0: k256 mul.rs:84-98 /
 * primeorder projective.rs:484-496) or lincomb_vartime (vartime=1: mul.rs:100-108 /
 * projective.rs:498-510), evaluated in chunks of `chunk` terms (0 = 4096) whose results are
 * added, because the reference keeps ~2 KB of tables per term. n == 0 gives
 * the identity. */
int ecref_msm(int curve, const uint8_t *scalars, const uint8_t *points_xy,
              const uint8_t *points_inf, size_t n, size_t chunk, int vartime,
              uint8_t *out_xy, uint8_t *out_inf);

/* out = a*G + b*P — MulByGeneratorVartime::mul_by_generator_and_mul_add_vartime
 * (k256 mul.rs:303-310; primeorder mul_backend.rs:29-40). */
int ecref_mul_base_and_mul_add_vartime(int curve, const uint8_t *a, const uint8_t *b,
                                       const uint8_t *p_xy, int p_inf, uint8_t *out_xy,
                                       uint8_t *out_inf);

/* ok[i] = ECDSA verification of (r_i, s_i) on digest integer z_i under public key Q_i — see ecref_ecdsa.c for the
 * algorithm and its provenance (ecdsa 0.17.0 hazmat::verify_prehashed, SEC1 4.1.4).  All fields L bytes big-endian,
 * q_xy affine x||y.  reject_high_s = the curve's NORMALIZE_S (k256/src/ecdsa.rs:104-106). */
int ecref_ecdsa_verify_batch(int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s,
                             const uint8_t *q_xy, size_t n, int reject_high_s, uint8_t *ok);

/* Public-key recovery: out_xy[i] = the key that (r_i, s_i) on digest integer z_i recovers to under recovery id byte
 * recid[i] (bit 0: y(R) odd, bit 1: x(R) = r + n), ok[i] = 1; or a zero record and ok[i] = 0 — ecdsa 0.17.0
 * `VerifyingKey::recover_from_prehash`, see ecref_ecdsa.c (reference vectors: k256/src/ecdsa.rs:170-262). */
int ecref_ecdsa_recover_batch(int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s,
                              const uint8_t *recid, size_t n, int reject_high_s, uint8_t *out_xy,
                              uint8_t *ok);

/* SM2DSA verification on the prehash (sm2/src/dsa/verifying.rs:138-171): e = SM3(ZA || M) as 32 bytes, (r, s), public key. */
int ecref_sm2dsa_verify_batch(const uint8_t *e, const uint8_t *r, const uint8_t *s, const uint8_t *q_xy, size_t n,
                              uint8_t *ok);
/* SM2DSA verification of messages: `VerifyingKey::new(distid, Q)?.verify(msg, sig)` — Z = SM3(ENTL || ID || a || b || G || Q)
 * (sm2/src/distid.rs:21-44), e = SM3(Z || M) (sm2/src/dsa/verifying.rs:126-130), then the prehash verification above.  One
 * identifier per batch, messages of one length, sigs = r || s.  ecref_sm3: the hash alone (test hook). */
int ecref_sm2dsa_verify_msg_batch(const uint8_t *distid, size_t distid_len, const uint8_t *q_xy,
                                  const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n,
                                  uint8_t *ok);
/* belt-hash (STB 34.101.31-2020 §7.8) and bign verification on the prehash / of messages (bignp256/src/ecdsa/verifying.rs:100-169);
 * all records little-endian: h 32 bytes, sigs 48 bytes S0 || S1, q_xy 64 bytes.  Pinned by bignp256/tests/ecdsa.rs:21-46. */
void ecref_belt_hash(const uint8_t *msg, size_t len, uint8_t out[32]);
int ecref_bign_verify_batch(const uint8_t *h, const uint8_t *sigs, const uint8_t *q_xy, size_t n, uint8_t *ok);
int ecref_bign_verify_msg_batch(const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n, uint8_t *ok);
int ecref_sm3(const uint8_t *msg, size_t len, uint8_t *out32);

/* ECDSA verification of messages: the curve's `DigestAlgorithm` digest (SHA-256 / 384 / 224 / 512), `bits2field`, then
 * ecref_ecdsa_verify_batch; sigs = r || s (2L bytes).  ecref_curve_digest: the digest alone (test hook). */
int ecref_ecdsa_verify_msg_batch(int curve, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len,
                                 const uint8_t *sigs, size_t n, int reject_high_s, uint8_t *ok);
int ecref_curve_digest(int curve, const uint8_t *msg, size_t len, uint8_t *out, size_t *out_len);

/* BIP340 verification over secp256k1 (k256/src/schnorr/verifying.rs:76-99) with the challenge hash e supplied by the
 * caller; see ecref_ecdsa.c. */
int ecref_schnorr_verify_batch(const uint8_t *e, const uint8_t *r, const uint8_t *s, const uint8_t *p_xy,
                               size_t n, uint8_t *ok);

/* The same from wire bytes: x-only keys (lifted with even y), messages of one length, 64-byte signatures; the challenge
 * hash is computed here (SHA-256).  See ecref_ecdsa.c. */
int ecref_schnorr_verify_raw_batch(const uint8_t *pk_x, const uint8_t *msgs, size_t msg_len,
                                   const uint8_t *sigs, size_t n, uint8_t *ok);

/* out[i] = (x_i, y_i) with y_i the square root of x^3 + a x + b of the requested parity —
 * `DecompressPoint::decompress(x_bytes, y_is_odd)` (primeorder/src/affine.rs:183-200, k256/src/arithmetic/affine.rs:261-280).
 * xs n*L bytes big-endian, y_is_odd n bytes; ok[i] = 0 and a zero record when x >= p or no root exists. */
int ecref_batch_decompress(int curve, const uint8_t *xs, const uint8_t *y_is_odd, size_t n,
                           uint8_t *out_xy, uint8_t *ok);

/* ---- smaller pieces, exposed so device-side code can be unit-checked against them ------- */

/* out = a (+,-,*) b mod p ; op: 0 add, 1 sub, 2 mul, 3 square(a), 4 invert(a) (0 -> 0),
 * 5 negate(a).  Canonical big-endian in and out. */
int ecref_field_op(int curve, int op, const uint8_t *a, const uint8_t *b, uint8_t *out);

/* Group law on affine-encoded operands through the reference's projective formulas:
 * op 0: P+Q (add_assign), 1: P+Q with Q taken as affine (add_assign_mixed), 2: 2P (double),
 * 3: -P.  (k256 projective.rs:96-217; primeorder point_arithmetic.rs:222-318) */
int ecref_point_op(int curve, int op, const uint8_t *p_xy, int p_inf, const uint8_t *q_xy,
                   int q_inf, uint8_t *out_xy, uint8_t *out_inf);

/* batch_normalize: n projective points (X||Y||Z, 3L bytes each, big-endian canonical) to
 * affine (k256 projective.rs:367-391; primeorder projective.rs:452-478). */
int ecref_batch_normalize(int curve, const uint8_t *xyz, size_t n, uint8_t *out_xy,
                          uint8_t *out_inf);

/* Radix16Decomposition::<D>::new over the low (D-1)/2 bytes of a big-endian scalar
 * (primeorder tables/radix16.rs:35-61).  digits must hold D entries. */
int ecref_radix16(const uint8_t *scalar_be, size_t scalar_len, int ndigits, int8_t *digits);

/* wnaf_form (wnaf/src/lib.rs:70-150) over little-endian bytes; returns the number of digits
 * written (<= bit_len+1), digits must hold bit_len+1 entries. */
int ecref_wnaf_form(const uint8_t *le_bytes, size_t nbytes, size_t bit_len, int window,
                    int8_t *digits);

/* k256 glv::decompose_scalar (mul/glv.rs:149-156): k -> (r1, r2) as canonical scalars mod n
 * (before sign folding). 32-byte big-endian each. */
int ecref_k256_glv_decompose(const uint8_t *k, uint8_t *r1, uint8_t *r2);

/* On-curve + range validation of n affine points (primeorder affine.rs:100-109). Returns
 * ECREF_OK or ECREF_ERR_POINT; *bad_index (may be NULL) receives the first offender. */
int ecref_validate_points(int curve, const uint8_t *points_xy, const uint8_t *points_inf,
                          size_t n, size_t *bad_index);

/* Scalar::reduce(bytes) for test generators: one conditional subtraction of n
 * (k256 scalar.rs:618-631; p256 scalar.rs:583-596; p384 scalar.rs:112-125). In place. */
int ecref_scalar_reduce(int curve, uint8_t *scalars, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ECREF_H */
