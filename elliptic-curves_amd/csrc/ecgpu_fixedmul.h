// ecgpu_fixedmul.h — one fixed-base scalar multiplication k*G over the signed comb table, the per-lane body of
// k_fixed_base (host + device; tests/hostcheck runs exactly this code on the CPU).
//
// Drop-in for `mul_by_generator` (k256/src/arithmetic/mul.rs:180-197; primeorder/src/tables/basepoint.rs:82-99).
// The reference walks 65 signed nibbles over a 33x8 projective table with full additions; here the scalar is
// folded to bits-1 bits (k G = -((n - k) G)), cut into nwin = (bits-1)/W + 1 signed W-bit windows, and window j
// selects entry |d_j| * 2^(W j) * G of an affine table: no doublings, nwin - 1 additions of affine points.  The sum is
// kept in XYZZ coordinates (first addition affine + affine, 4M + 2S; the others 8M + 2S, against 11M + constants for
// the complete mixed addition), the sign of a digit is folded into the addition formula.
//
// The XYZZ additions are incomplete (accumulator = +-entry is not handled); that case cannot occur.  With signed
// digits |d_i| <= 2^(W-1) the partial sum S_j = sum_{i<j} d_i 2^(W i) satisfies |S_j| < 2^(W j) (geometric sum), the
// entry added next is e_j = d_j 2^(W j) with |e_j| >= 2^(W j), so S_j != +-e_j as integers; as group elements they
// coincide only if S_j -+ e_j = 0 (mod n).  For j < nwin - 1: |S_j| + |e_j| <= 2^(W (j+1)) <= 2^(bits-1) < n.
// For the top window S_j + e_j = k (folded, 0 < k < n/2), so S_j = -e_j is impossible, and S_j = e_j (mod n) means
// 2 S_j - k = -n (the only multiple of n in range), i.e. |S_j| > n/4: that needs W (nwin - 1) = bits - 1, where the
// top digit is the carry alone (e_j = 2^(bits-1), S_j = k - 2^(bits-1)) and the condition reads k = 2^bits - n —
// a value < 2^(bits - 31) for the curves whose n is within 2^-4 of 2^bits (k256: 2^129, p256 / sm2: 2^225, p384: 2^190,
// p224: 2^113, p192: 2^96), so with W <= 26 the window below the top one is zero and hands no carry up.  Contradiction.
// The other curves (brainpoolP256r1: n = 0.66 * 2^256; p521: 521 bits in 544) do not rely on this: COMB_NEEDS_CHECK.
// tests: comb_corner_scalars (tests/gpu_common.py) at W = 5 and 15 (top window at bit 255) and at the default widths.
#pragma once

#include "ecgpu_point.h"
#include "ecgpu_recode.h"

namespace ecgpu {

template <class C, class Table>
ECGPU_HD Affine<C> load_entry(const Table& table, int window, uint32_t index) {
    PackedPoint<2 * C::N> pw;
    table.load(pw, window, index);
    Affine<C> q;
    q.x = Field<C>::unpack(pw.w).e;
    q.y = Field<C>::unpack(pw.w + C::N).e;
    return q;
}

// Table: void load(PackedPoint<2N>&, int window, uint32_t index) const   — entry (index + 1) * 2^(W * window) * G,
// packed storage form.  (Requesting the entry of window j + 1 before the addition of window j was measured and is 4 %
// slower: the 16 extra live registers cost the k256 kernel two of its four waves per SIMD, and occupancy hides the
// gather latency better than software prefetching does.  An LDS-DMA prefetch (`global_load_lds_dwordx4` of the next
// entry under the current addition, no staging registers, 3 waves per SIMD) was measured too: 0.644 ms against
// 0.643 ms — the kernel is not waiting for its gathers.  Repeated with the XYZZ kernel, where the staging registers fit
// without costing a wave (168 VGPRs, 3 waves per SIMD): 0.545 ms against 0.52 ms.  And once more in round 5 in the form that helped the
// MSM's accumulation loop — the next entry requested unconditionally behind the pinned unpacking of the current one, no copies at
// the back edge, 168 VGPRs: kernel 0.511 / 0.517 against 0.504 / 0.510 ms, same box, alternating — profiles/r05/fixed_prefetch_ab.txt.)
// whether the never-exceptional argument for the comb does not cover the curve: n not within 2^-4 of 2^bits
template <class C>
constexpr bool COMB_NEEDS_CHECK = C::ORDER[C::N - 1] < 0xF0000000u;

template <class C, class Table>
ECGPU_HD Proj<C> fixed_base_mul(const uint32_t* k_in, const Table& table, int w, int nwin, const Fe<C::NL>& b) {
    using G = Group<C>;
    constexpr int N = C::N;
    uint32_t k[N];
#pragma unroll
    for (int i = 0; i < N; i++) k[i] = k_in[i];
    const bool flip = fold_scalar<N>(k, C::ORDER);
    uint32_t carry = 0;
    // state 0: nothing added yet; 1: acc = (x, y) affine (one entry); 2: acc in XYZZ coordinates
    int state = 0;
    Xyzz<C> acc;
    acc.x = acc.y = acc.zz = acc.zzz = Field<C>::one().e;
#pragma unroll 1
    for (int j = 0; j < nwin; j++) {
        int d = signed_window_step(get_bits<N>(k, j * w, w), w, &carry);
        if (d != 0) {
            Affine<C> q = load_entry<C>(table, j, (uint32_t)(d < 0 ? -d : d) - 1);
            const bool neg = (d < 0) != flip;
            if (state == 2) {
                acc = G::xyzz_madd(acc, q, neg);
            } else if (state == 1) {
                Affine<C> a;
                a.x = acc.x;
                a.y = acc.y;
                acc = G::xyzz_mmadd(a, q, neg);
                state = 2;
            } else {
                if (neg) q.y = G::neg_coord(q.y);
                acc.x = q.x;
                acc.y = q.y;
                state = 1;
            }
        }
    }
    if (state == 0) return G::identity();
    if (state == 1) {
        Affine<C> a;
        a.x = acc.x;
        a.y = acc.y;
        return G::from_affine(a);
    }
    if constexpr (COMB_NEEDS_CHECK<C>) {
        // the argument above needs n close to 2^bits (2^bits - n small).  For brainpoolP256r1 (n = 0.66 * 2^256) the scalar
        // k = 2^256 - n is not folded and, at the widths with W (nwin - 1) = 255, meets acc = entry at the top window:
        // an exceptional addition zeroes ZZ for good, so ZZ == 0 is the exact test; the scalar is then redone with
        // complete additions.
        if (Field<C>::is_zero(G::mj(acc.zz))) {
            uint32_t carry2 = 0;
            Proj<C> r = G::identity();
#pragma unroll 1
            for (int j = 0; j < nwin; j++) {
                int d = signed_window_step(get_bits<N>(k, j * w, w), w, &carry2);
                if (d != 0) {
                    Affine<C> q = load_entry<C>(table, j, (uint32_t)(d < 0 ? -d : d) - 1);
                    r = G::add_mixed(r, q, b, (d < 0) != flip);
                }
            }
            return r;
        }
    }
    return G::xyzz_to_proj(acc);
}


}  // namespace ecgpu
