// ecgpu_fixedmul.h — one fixed-base scalar multiplication k*G over the signed comb table, the per-lane body of
// k_fixed_base (host + device; tests/hostcheck runs exactly this code on the CPU).
//
// Drop-in for `mul_by_generator` (k256/src/arithmetic/mul.rs:180-197; primeorder/src/tables/basepoint.rs:82-99).
// The reference walks 65 signed nibbles over a 33x8 projective table with full additions; here the scalar is
// folded to bits-1 bits (k G = -((n - k) G)), cut into nwin = (bits-1)/W + 1 signed W-bit windows, and window j
// selects entry |d_j| * 2^(W j) * G of an affine table: nwin - 1 complete *mixed* additions (RCB Alg 8 / Alg 5),
// no doublings, no exceptional cases.  The first window initialises the accumulator (an addition to the identity
// would be a 2000-instruction copy), the sign of a digit is folded into the addition formula.
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
// 0.643 ms — the kernel is not waiting for its gathers.)
template <class C, class Table>
ECGPU_HD Proj<C> fixed_base_mul(const uint32_t* k_in, const Table& table, int w, int nwin, const Fe<C::NL>& b) {
    using G = Group<C>;
    constexpr int N = C::N;
    uint32_t k[N];
#pragma unroll
    for (int i = 0; i < N; i++) k[i] = k_in[i];
    const bool flip = fold_scalar<N>(k, C::ORDER);
    uint32_t carry = 0;
    Proj<C> acc = G::identity();
    {
        int d = signed_window_step(get_bits<N>(k, 0, w), w, &carry);
        if (d != 0) {
            Affine<C> q = load_entry<C>(table, 0, (uint32_t)(d < 0 ? -d : d) - 1);
            if ((d < 0) != flip) q.y = G::neg_coord(q.y);
            acc = G::from_affine(q);
        }
    }
#pragma unroll 1
    for (int j = 1; j < nwin; j++) {
        int d = signed_window_step(get_bits<N>(k, j * w, w), w, &carry);
        if (d != 0) {
            Affine<C> q = load_entry<C>(table, j, (uint32_t)(d < 0 ? -d : d) - 1);
            acc = G::add_mixed(acc, q, b, (d < 0) != flip);
        }
    }
    return acc;
}


}  // namespace ecgpu
