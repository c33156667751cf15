// ecgpu_ecdsa.h — batch ECDSA verification around the aG + bP kernels (HIP only).
//
// the dominant caller of `mul_by_generator_and_mul_add_vartime`
// (primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310).  The verification equation itself
// lives in the un-vendored `ecdsa` crate 0.17.0 (Cargo.lock:428-429, `hazmat::verify_prehashed`, instantiated at
// p256/src/ecdsa.rs:69,161-169, p384/src/ecdsa.rs:179-187, k256/src/ecdsa.rs:104-106); it is the published
// algorithm of SEC1 v2 §4.1.4 / FIPS 186-5 §6.4.2:
//     z = the leftmost bits of the digest as an integer, reduced mod n        (`Reduce<FieldBytes>`)
//     reject unless 1 <= r, s < n   (and, where the curve sets NORMALIZE_S, unless s <= (n-1)/2)
//     w = s^-1 mod n,  u1 = z w,  u2 = r w,  R = u1 G + u2 Q;  accept iff R != identity and x(R) mod n == r
// One lane per signature; an invalid element never fails the batch, it just gets ok = 0.
#pragma once

#include "ecgpu_kernels.h"
#include "ecgpu_scalar.h"
#include "ecgpu_sha256.h"
#include "ecgpu_hash.h"
#include "ecgpu_sm3.h"
#include "ecgpu_belt.h"
#include "ecgpu_verify.h"

namespace ecgpu {

// ---- inverses modulo the group order for a whole batch: Montgomery's trick, one division-step inversion per LANE instead of one per
// signature.  `Scalar::invert` (k256/src/arithmetic/scalar.rs:139-143; primefield/src/monty.rs:373-375) is what verification
// (s^-1) and recovery (r^-1) call once per signature: 21.5 k dependent instructions each, 90 % of k_ecdsa_prepare.  Lane t owns
// elements t, t + T, t + 2T, ... (a wave touches consecutive records, like k_normalize): running products in the Montgomery domain
// (one multiplication per product), the prefix products parked in `prefix` (N words per element), one inversion of the lane's
// product, a backward pass that peels the inverses off.  An element outside [1, n - 1] is left out of the product and gets 0 —
// the prepare kernels fail it on their range check and never use its inverse.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_scalar_batch_inv(const uint8_t* __restrict__ in, size_t n, size_t nthreads, uint32_t* __restrict__ prefix, uint8_t* __restrict__ out) {
    using S = ScalarN<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads || t >= n) return;
    uint32_t acc[N], one_m[N];
    {
        uint32_t one[N];
#pragma unroll
        for (int i = 0; i < N; i++) one[i] = i == 0 ? 1u : 0u;
        S::to_mont(one_m, one);                                    // R mod n
    }
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = one_m[i];
    for (size_t j = t; j < n; j += nthreads) {
        uint32_t a[N], am[N];
        load_wire<C>(a, in + j * WB);
        const bool ok = !S::is_zero(a) && S::in_range(a);
        uint32_t* pj = prefix + j * N;
#pragma unroll
        for (int i = 0; i < N; i++) pj[i] = acc[i];                // the product of the lane's earlier elements (Montgomery form)
        if (ok) {
            S::to_mont(am, a);
            S::mont_mul(acc, acc, am);
        }
    }
    uint32_t inv[N];
    {
        // acc = P R as an integer; x = acc^-1 = P^-1 R^-1; x R2 / R = P^-1; P^-1 R2 / R = P^-1 R: the Montgomery form of P^-1
        uint32_t x[N];
        S::inv(x, acc);
        S::to_mont(x, x);
        S::to_mont(inv, x);
    }
    const size_t last = t + ((n - 1 - t) / nthreads) * nthreads;
    for (size_t j = last;; j -= nthreads) {
        uint32_t a[N], am[N], w[N];
        load_wire<C>(a, in + j * WB);
        const bool ok = !S::is_zero(a) && S::in_range(a);
#pragma unroll
        for (int i = 0; i < N; i++) w[i] = 0u;
        if (ok) {
            uint32_t pre[N], wm[N];
            const uint32_t* pj = prefix + j * N;
#pragma unroll
            for (int i = 0; i < N; i++) pre[i] = pj[i];
            S::mont_mul(wm, inv, pre);                             // (a_0 .. a_j)^-1 (a_0 .. a_(j-1)) = a_j^-1
            S::to_mont(am, a);
            S::mont_mul(inv, inv, am);                             // (a_0 .. a_(j-1))^-1
            S::from_mont(w, wm);
        }
        store_wire<C>(out + j * WB, w);
        if (j < nthreads) break;
    }
}

// prepare: range checks, public-key validation, u1 / u2 as wire-format scalars for the scalar-mul kernels.
// Elements that are already known to fail get u1 = u2 = 0 and Q = G so that the arithmetic kernels see valid input.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_prepare(const uint8_t* __restrict__ z, const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                const uint8_t* __restrict__ q_xy, size_t n, int reject_high_s, uint8_t* __restrict__ u1_out,
                uint8_t* __restrict__ u2_out, uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid,
                const uint8_t* __restrict__ s_inv) {           // s_inv: the batch's s^-1 (k_scalar_batch_inv), or null
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t zw[N], rw[N], sw[N], cx[N], cy[N], u1[N], u2[N], w[N];
    load_wire<C>(zw, z + i * WB);
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    load_wire<C>(cx, q_xy + i * (2 * WB));
    load_wire<C>(cy, q_xy + i * (2 * WB) + WB);
    if (s_inv) load_wire<C>(w, s_inv + i * WB);
    const bool ok = ecdsa_prepare_words<C>(zw, rw, sw, cx, cy, reject_high_s, u1, u2, s_inv ? w : nullptr);      // ecgpu_verify.h
    store_wire<C>(u1_out + i * WB, u1);
    store_wire<C>(u2_out + i * WB, u2);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}

// finish: x(R) mod n == r
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_finish(const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf, const uint8_t* __restrict__ r,
               const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[N], rw[N];
    load_wire<C>(x, r_xy + i * (2 * WB));
    load_wire<C>(rw, r + i * WB);
    const bool eq = ecdsa_finish_words<C>(x, rw);
    ok_out[i] = (valid[i] && !r_inf[i] && eq) ? 1 : 0;
}

// ECDSA verification of MESSAGES — `Verifier::verify(msg, &signature)` of `ecdsa::VerifyingKey<C>`: the curve's digest
// (`DigestAlgorithm`, EcdsaDigest<C> in ecgpu_hash.h) on the device, z = bits2field(digest) (ecdsa `hazmat::bits2field`: the
// leftmost L bytes, left-padded with zeros when the digest is shorter: p521 with SHA-512), r and s split out of the 2L-byte
// signatures; k_ecdsa_prepare and the rest as for the prehash entry point.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_hash_msg(const uint8_t* __restrict__ msgs, size_t msg_len, const uint8_t* __restrict__ sigs, size_t n,
                 uint8_t* __restrict__ z_out, uint8_t* __restrict__ r_out, uint8_t* __restrict__ s_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value, D = EcdsaDigest<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (D != 0) {
        uint8_t digest[D];
        const HashPiece one[1] = {{msgs + i * msg_len, msg_len}};
        sha2_pieces<D, 1>(digest, one);
        uint32_t zw[N];
#pragma unroll
        for (int j = 0; j < N; j++) zw[j] = 0;
        constexpr int TAKE = D < WB ? D : WB;                       // digest bytes used: the leftmost TAKE
#pragma unroll
        for (int j = 0; j < TAKE; j++) {
            const int pos = TAKE - 1 - j;                           // significance of digest byte j within the integer
            zw[pos / 4] |= (uint32_t)digest[j] << (8 * (pos % 4));
        }
        store_wire<C>(z_out + i * WB, zw);
        uint32_t w[N];
        load_wire<C>(w, sigs + i * (2 * WB));
        store_wire<C>(r_out + i * WB, w);
        load_wire<C>(w, sigs + i * (2 * WB) + WB);
        store_wire<C>(s_out + i * WB, w);
    }
}

// ---- ECDSA public-key recovery: ecdsa 0.17.0 `VerifyingKey::recover_from_prehash` (see ecgpu_verify.h) -----------------------
//     prepare: checks, R = decompress(r or r + n, parity), a = -(z / r), b = s / r;  then a G + b R by the kernels of
//     ecgpu_batch_mul_base_and_mul_add;  finish: the key, or a zero record and ok = 0 (failed checks, or the identity,
//     which `VerifyingKey::from_affine` rejects)
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_recover_prepare(const uint8_t* __restrict__ z, const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                        const uint8_t* __restrict__ recid, size_t n, int reject_high_s, uint8_t* __restrict__ a_out,
                        uint8_t* __restrict__ b_out, uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid,
                        const uint8_t* __restrict__ r_inv) {       // r_inv: the batch's r^-1 (k_scalar_batch_inv), or null
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t zw[N], rw[N], sw[N], cx[N], cy[N], a[N], b[N], w[N];
    load_wire<C>(zw, z + i * WB);
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    if (r_inv) load_wire<C>(w, r_inv + i * WB);
    const bool ok = ecdsa_recover_prepare_words<C>(zw, rw, sw, recid[i], reject_high_s, a, b, cx, cy, r_inv ? w : nullptr);   // ecgpu_verify.h
    store_wire<C>(a_out + i * WB, a);
    store_wire<C>(b_out + i * WB, b);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}
// in place on the normalised sums: failed elements become zero records
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_recover_finish(uint8_t* __restrict__ xy, const uint8_t* __restrict__ inf, const uint8_t* __restrict__ valid, size_t n,
                       uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool ok = valid[i] && !inf[i];
    if (!ok) {
        uint32_t zero[N];
#pragma unroll
        for (int j = 0; j < N; j++) zero[j] = 0;
        store_wire<C>(xy + i * (2 * WB), zero);
        store_wire<C>(xy + i * (2 * WB) + WB, zero);
    }
    ok_out[i] = ok ? 1 : 0;
}

// ---- SM2DSA verification on the prehash: sm2/src/dsa/verifying.rs:138-171 ----------------------------------------------------
//     e = SM3(ZA || M) as 32 bytes (computed by the caller: ZA depends on the signer's identity), reduced mod n
//     reject unless 1 <= r, s < n;  t = r + s mod n, reject t = 0;  (x1, y1) = s G + t Q;  accept iff r == e + x1 mod n
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_sm2dsa_prepare(const uint8_t* __restrict__ r, const uint8_t* __restrict__ s, const uint8_t* __restrict__ q_xy, size_t n,
                 uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out, uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[N], sw[N], cx[N], cy[N], t[N];
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    load_wire<C>(cx, q_xy + i * (2 * WB));
    load_wire<C>(cy, q_xy + i * (2 * WB) + WB);
    const bool ok = sm2dsa_prepare_words<C>(rw, sw, cx, cy, t);                               // ecgpu_verify.h
    store_wire<C>(a_out + i * WB, sw);
    store_wire<C>(b_out + i * WB, t);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_sm2dsa_finish(const uint8_t* __restrict__ e, const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf,
                const uint8_t* __restrict__ r, const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ew[N], x[N], rw[N];
    load_wire<C>(ew, e + i * WB);
    load_wire<C>(x, r_xy + i * (2 * WB));
    load_wire<C>(rw, r + i * WB);
    ok_out[i] = (valid[i] && sm2dsa_finish_words<C>(ew, x, r_inf[i] != 0, rw)) ? 1 : 0;
}

// SM2DSA verification of MESSAGES — `VerifyingKey::new(distid, Q)?.verify(msg, sig)`: the identity hash
// Z = SM3(ENTL || ID || a || b || xG || yG || xA || yA) (`hash_z`, sm2/src/distid.rs:21-44) and e = SM3(Z || M) (`hash_msg`,
// sm2/src/dsa/verifying.rs:126-130) on the device (ecgpu_sm3.h); r and s are split out of the 64-byte signatures, and the
// prehash kernels above do the rest.  A key with a coordinate >= p hashes as given and is rejected by k_sm2dsa_prepare.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_sm2dsa_hash_msg(const uint8_t* __restrict__ distid, size_t distid_len, const uint8_t* __restrict__ q_xy,
                  const uint8_t* __restrict__ msgs, size_t msg_len, const uint8_t* __restrict__ sigs, size_t n,
                  uint8_t* __restrict__ e_out, uint8_t* __restrict__ r_out, uint8_t* __restrict__ s_out) {
    constexpr int N = C::N;
    static_assert(N == 8, "SM2DSA is defined over the 256-bit sm2 curve");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[N], sw[N], ew[N];
    load_be_vec<N>(rw, sigs + i * 64);
    load_be_vec<N>(sw, sigs + i * 64 + 32);
    Sm3::sm2_message_hash<C>(ew, distid, distid_len, q_xy + i * 64, msgs + i * msg_len, msg_len);
    store_be_vec<N>(e_out + i * 32, ew);
    store_be_vec<N>(r_out + i * 32, rw);
    store_be_vec<N>(s_out + i * 32, sw);
}

// ---- bign verification on the prehash: bignp256/src/ecdsa/verifying.rs:100-147 (STB 34.101.45-2013 §7.2) ---------------------------
//     the 48-byte signature is S0 (16 bytes) || S1 (32 bytes), little-endian; reject S0 = 0, S1 = 0, S1 >= q (`Signature::from_bytes`)
//     R = ((S1 + H) mod q) G + (S0 + 2^128) Q;  reject R = O;  t = the first 16 bytes of belt-hash(OID(h) || <R>_2l || H);
//     accept iff S0 == t.   <R>_2l is the x coordinate of R as 32 little-endian bytes: the bign256 wire record.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_bign_prepare(const uint8_t* __restrict__ h, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ q_xy, size_t n,
               uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out, uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    static_assert(N == 8 && WB == 32 && WireLe<C>::value, "bign-curve256v1: 32-byte little-endian records");
    static_assert(BLOCK == 256, "k_bign_finish / k_bign_hash_msg stage the 256-byte S-box with one byte per lane");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t hw[N], s0w[4], s1w[N], cx[N], cy[N], a[N], b[N];
    load_wire<C>(hw, h + i * WB);
    load_words_vec<4>(s0w, reinterpret_cast<const uint32_t*>(sigs + i * 48));
    load_wire<C>(s1w, sigs + i * 48 + 16);
    load_wire<C>(cx, q_xy + i * (2 * WB));
    load_wire<C>(cy, q_xy + i * (2 * WB) + WB);
    const bool ok = bign_prepare_words<C>(hw, s0w, s1w, cx, cy, a, b);                          // ecgpu_verify.h
    store_wire<C>(a_out + i * WB, a);
    store_wire<C>(b_out + i * WB, b);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_bign_finish(const uint8_t* __restrict__ h, const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf,
              const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    constexpr int WB = WireBytes<C>::value;
    __shared__ uint8_t sbox[256];                                   // the S-box where byte-indexed lookups are cheap
    sbox[threadIdx.x] = Belt::H[threadIdx.x];
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s0w[4], t[8];
    load_words_vec<4>(s0w, reinterpret_cast<const uint32_t*>(sigs + i * 48));
    const HashPiece pc[3] = {{Belt::OID, sizeof(Belt::OID)}, {r_xy + i * (2 * WB), (size_t)WB}, {h + i * WB, (size_t)WB}};
    Belt::hash_pieces<3>(sbox, t, pc);
    const bool eq = ((t[0] ^ s0w[0]) | (t[1] ^ s0w[1]) | (t[2] ^ s0w[2]) | (t[3] ^ s0w[3])) == 0u;
    ok_out[i] = (valid[i] && r_inf[i] == 0 && eq) ? 1 : 0;
}
// H = belt-hash(message) per element (`hash_msg`, bignp256/src/ecdsa/verifying.rs:87-91); the prehash kernels do the rest
static __global__ void __launch_bounds__(BLOCK)
k_bign_hash_msg(const uint8_t* __restrict__ msgs, size_t msg_len, size_t n, uint8_t* __restrict__ h_out) {
    __shared__ uint8_t sbox[256];
    sbox[threadIdx.x] = Belt::H[threadIdx.x];
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t hw[8];
    const HashPiece pc[1] = {{msgs + i * msg_len, msg_len}};
    Belt::hash_pieces<1>(sbox, hw, pc);
    store_words_vec<8>(reinterpret_cast<uint32_t*>(h_out + i * 32), hw);
}

// ---- Schnorr (BIP340) verification: k256/src/schnorr/verifying.rs:76-99 ---------------------------------------------
//     e = tagged_hash("BIP0340/challenge", r || pk || m) reduced mod n   (computed by the caller; reduced here)
//     R = s*G + (-e)*P  via mul_by_generator_and_mul_add_vartime;  accept iff R != identity, y(R) even, x(R) == r
// Signature parsing (k256/src/schnorr.rs:132-150): r < p, 0 < s < n.  P is the verifying key's stored affine point
// (lifted with even y by VerifyingKey::from_bytes — ecgpu_batch_decompress with y_is_odd = 0).
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_prepare(const uint8_t* __restrict__ e, const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                  const uint8_t* __restrict__ p_xy, size_t n, uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out,
                  uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ew[N], rw[N], sw[N], cx[N], cy[N];
    load_wire<C>(ew, e + i * WB);
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    load_wire<C>(cx, p_xy + i * (2 * WB));
    load_wire<C>(cy, p_xy + i * (2 * WB) + WB);
    uint32_t ne[N];
    const bool ok = schnorr_prepare_words<C>(ew, rw, sw, cx, cy, ne);                       // ecgpu_verify.h
    store_wire<C>(a_out + i * WB, sw);
    store_wire<C>(b_out + i * WB, ne);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}

// The whole of `VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)` (k256/src/schnorr/verifying.rs:76-99,149-160) from
// wire bytes: lift_x of the 32-byte key (even y; fails for x >= p or a non-residue), signature parsing, the challenge
// hash e = tagged_hash("BIP0340/challenge", r || pk || msg) on the device, then the same a = s, b = -e as above.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_prepare_raw(const uint8_t* __restrict__ pk_x, const uint8_t* __restrict__ msgs, size_t msg_len,
                      const uint8_t* __restrict__ sigs, size_t n, uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out,
                      uint8_t* __restrict__ q_out, uint8_t* __restrict__ r_out, uint8_t* __restrict__ valid) {
    using S = ScalarN<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    static_assert(N == 8 && C::A_IS_ZERO, "BIP340 is defined over secp256k1");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[N], sw[N], cx[N], cy[N];
    load_be_vec<N>(cx, pk_x + i * 32);
    load_be_vec<N>(rw, sigs + i * 64);
    load_be_vec<N>(sw, sigs + i * 64 + 32);
    bool ok = !mp_geq<N>(rw, C::P) && !S::is_zero(sw) && S::in_range(sw);
    ok = schnorr_lift_x<C>(cx, cy) && ok;                                                   // ecgpu_verify.h
    uint32_t ew[N], ne[N];
    Sha256::bip340_challenge(ew, sigs + i * 64, pk_x + i * 32, msgs + i * msg_len, msg_len);
    schnorr_neg_challenge<C>(ne, ew);
    verify_blank<C>(ok, sw, ne, cx, cy);
    store_be_vec<N>(a_out + i * 32, sw);
    store_be_vec<N>(b_out + i * 32, ne);
    store_be_vec<N>(q_out + i * 64, cx);
    store_be_vec<N>(q_out + i * 64 + 32, cy);
    store_be_vec<N>(r_out + i * 32, rw);
    valid[i] = ok ? 1 : 0;
}

template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_finish(const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf, const uint8_t* __restrict__ r,
                 const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[N], y[N], rw[N];
    load_wire<C>(x, r_xy + i * (2 * WB));
    load_wire<C>(y, r_xy + i * (2 * WB) + WB);
    load_wire<C>(rw, r + i * WB);
    ok_out[i] = (valid[i] && !r_inf[i] && schnorr_finish_words<C>(x, y, rw)) ? 1 : 0;
}

// ---- ECDH: x-coordinate of k*P -----------------------------------------------------------------------------------------
// `elliptic_curve::ecdh::diffie_hellman` (elliptic-curve 0.14.1, un-vendored; used through k256/src/ecdh.rs,
// p256/src/ecdh.rs, p384/src/ecdh.rs): SharedSecret = x((public * secret).to_affine()).  The scalar multiplication is
// k_var_base; this kernel keeps x and reports ok = 0 for an identity result (only k = 0 can produce one: the
// reference's NonZeroScalar / PublicKey types exclude it).
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_extract_x(const uint8_t* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n, uint8_t* __restrict__ out_x,
            uint8_t* __restrict__ ok) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    copy_wire<C>(out_x + i * WB, xy + i * (2 * WB));
    ok[i] = inf[i] ? 0 : 1;
}

// ---- point decompression: DecompressPoint::decompress(x_bytes, y_is_odd) -----------------------------------------------
// primeorder/src/affine.rs:183-200, k256/src/arithmetic/affine.rs:261-280: alpha = x^3 + a x + b,
// beta = sqrt(alpha), y = beta or -beta by the parity of the canonical value.  ok = 0 (zero record) for x >= p or a
// non-residue.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_decompress(const uint8_t* __restrict__ xs, const uint8_t* __restrict__ y_is_odd, size_t n, uint8_t* __restrict__ out_xy,
             uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t cx[N], cy[N];
    load_wire<C>(cx, xs + i * WB);
    const bool ok = decompress_words<C>(cx, y_is_odd[i] != 0, cy);                          // ecgpu_verify.h
    store_wire<C>(out_xy + i * (2 * WB), cx);
    store_wire<C>(out_xy + i * (2 * WB) + WB, cy);
    ok_out[i] = ok ? 1 : 0;
}

// SEC1-compressed points INTO the scalar-multiplication entry points (ecgpu_msm_compressed, ecgpu_batch_mul_compressed): record i
// is x (one wire element) + a tag byte — 0x02 / 0x03: the point with that x and even / odd y (`FromSec1Point` ->
// `DecompressPoint::decompress`, primeorder/src/affine.rs:183-200,352-366; k256/src/arithmetic/affine.rs:261-280), 0x00: the
// identity (`Sec1Point::identity`, the one-byte encoding).  out_xy / out_inf are the x || y + flag records the path takes.  Any
// other tag, x >= p, or an x with no point on the curve raises ST_BAD_POINT — the reference's `CtOption::None`, which ends a
// `lincomb` over decoded keys before it starts.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_decompress_tagged(const uint8_t* __restrict__ xs, const uint8_t* __restrict__ tags, size_t n, uint8_t* __restrict__ out_xy,
                    uint8_t* __restrict__ out_inf, int* status) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t tag = tags[i];
    uint32_t cx[N], cy[N];
    load_wire<C>(cx, xs + i * WB);
    bool ok;
    if (tag == 0u) {                                        // the identity: x must be absent / zero; a zero record goes on
        ok = mp_is_zero<N>(cx);
#pragma unroll
        for (int k = 0; k < N; k++) cx[k] = cy[k] = 0;
    } else {
        ok = (tag == 2u || tag == 3u) && decompress_words<C>(cx, tag == 3u, cy);          // ecgpu_verify.h
    }
    if (!ok) atomicOr(status, ST_BAD_POINT);
    store_wire<C>(out_xy + i * (2 * WB), cx);
    store_wire<C>(out_xy + i * (2 * WB) + WB, cy);
    out_inf[i] = tag == 0u ? 1 : 0;
}

}  // namespace ecgpu
