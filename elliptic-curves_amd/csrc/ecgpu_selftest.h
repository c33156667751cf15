// ecgpu_selftest.h — device-side known-answer kernels for the arithmetic every other kernel is built from.
//
// tests/hostcheck runs the same __host__ __device__ field / group code on the CPU (g++ build); these two kernels run it
// as gfx950 code, one lane per element, so that the -m gpu tests can compare the DEVICE's field and point operations with
// the oracle and the reference's own field vectors (k256/src/test_vectors/field.rs) directly, not only through
// scalar-multiplication results.  Operation codes follow tests/hostcheck (field_op / point_op).
#pragma once

#include "ecgpu_kernels.h"
#include "ecgpu_rows.h"

namespace ecgpu {

// field: 0 a + b, 1 a - b, 2 a * b, 3 a^2, 4 1/a (division steps; 0 -> 0), 5 -a, 7 2a, 8 pack/unpack of the lazy value
// 2a + b, 9 the fused a*b - (a + b)*b, 15 (k256) 7 a * b + 6 a * b with BOTH operands lazy at the largest limb magnitudes a product
// takes (7 x 1 and 3 x 2 + 1 x 1), reduced by the assembly blocks AND by the compiler's k_reduce from the same columns: the nine
// limbs must agree, else the record is all ones; 10 1/a by Fermat, 11 sqrt(a) or 0, 12 a 25-step chain of lazily reduced
// operations at the magnitudes the point formulas use, 20 a through the wire -> words -> wire conversion only, 21 a through
// the internal domain and back; 16 (k256; n a multiple of 64) the ROW-PARALLEL multiplication of ecgpu_rows.h: the operands of the
// wave's first four lanes on its four rows, 7 a * b + 3 a * 2 b = 13 a b of lane (i mod 4) of the wave in every lane i.
// 17 1/a by the variable-time division steps (0 -> 0).  Inputs must be canonical (< p), else ST_BAD_POINT.
template <class C>
__global__ void __launch_bounds__(BLOCK) k_selftest_field(int op, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
                                                          uint8_t* __restrict__ out, int* status) {
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wa[N], wb[N], wr[N];
    load_wire<C>(wa, a + i * WB);
    if (b) load_wire<C>(wb, b + i * WB);
    else {
#pragma unroll
        for (int t = 0; t < N; t++) wb[t] = 0;
    }
    if (mp_geq<N>(wa, C::P) || mp_geq<N>(wb, C::P)) atomicOr(status, ST_BAD_POINT);
    typename F::M1 x = F::from_canonical(wa), y = F::from_canonical(wb);
    switch (op) {
    case 0: F::to_canonical(wr, F::add(x, y)); break;
    case 1: F::to_canonical(wr, F::norm(F::sub(x, y))); break;
    case 2: F::to_canonical(wr, F::mul(x, y)); break;
    case 3: F::to_canonical(wr, F::sqr(x)); break;
    case 4: F::to_canonical(wr, F::inv(x)); break;
    case 17: F::to_canonical(wr, F::template inv<true>(x)); break;          // the variable-time division steps (ModInv::invert_var)
    case 5: F::to_canonical(wr, F::neg(x)); break;
    case 7: F::to_canonical(wr, F::dbl(x)); break;
    case 8: {
        uint32_t w[N];
        F::pack(w, F::norm(F::add(F::dbl(x), y)));
        F::to_canonical(wr, F::unpack(w));
        break;
    }
    case 9: F::to_canonical(wr, F::mul2(x, y, F::add(x, y), F::neg(y))); break;
    case 10: F::to_canonical(wr, F::inv_fermat(x)); break;
    case 13:                                   // x y - (2 x + y): k256 through the fused F::mul_sub, the others in two steps
        if constexpr (C::REPR == REPR_U29_K256) F::to_canonical(wr, F::mul_sub(x, y, F::add(F::dbl(x), y)));
        else F::to_canonical(wr, F::norm(F::sub(F::mul(x, y), F::add(F::dbl(x), y))));
        break;
    case 14:                                   // (x + y)^2 - 5 y: F::sqr_sub with a lazy operand and the largest subtrahend in use
        if constexpr (C::REPR == REPR_U29_K256) F::to_canonical(wr, F::sqr_sub(F::add(x, y), F::add(y, F::dbl(F::dbl(y)))));
        else F::to_canonical(wr, F::norm(F::sub(F::sqr(F::norm(F::add(x, y))), F::add(y, F::dbl(F::dbl(y))))));
        break;
    case 15:
        if constexpr (C::REPR == REPR_U29_K256) {
            const auto x2 = F::dbl(x);                                  // limb magnitudes 2, 3, 7
            const auto x3 = F::add(x2, x);
            const auto x7 = F::add(F::add(x3, x3), x);
            const auto y2 = F::dbl(y);
            uint64_t c1[17], c2[17];
            bool same = true;
            // 7 x * y: one product at the limit of a column (magnitude product 7)
            F::k_columns(c1, x7.e.v, y.e.v, false);
#pragma unroll
            for (int t = 0; t < 17; t++) c2[t] = c1[t];
            const auto r1 = F::k_reduce(c1), r2 = F::k_reduce_cpp(c2);
            // 3 x * 2 y + x * y = 7 x y as two products under one reduction (mul2's shape: 6 + 1)
            F::k_columns(c1, x3.e.v, y2.e.v, false);
            F::k_columns(c1, x.e.v, y.e.v, true);
#pragma unroll
            for (int t = 0; t < 17; t++) c2[t] = c1[t];
            const auto r3 = F::k_reduce(c1), r4 = F::k_reduce_cpp(c2);
#pragma unroll
            for (int t = 0; t < 9; t++) same = same && r1.v[t] == r2.v[t] && r3.v[t] == r4.v[t];
            F::to_canonical(wr, F::add(F::template wrap<1, 1>(r1), F::template wrap<1, 1>(r3)));                 // 14 x y
            if (!same) {
#pragma unroll
                for (int t = 0; t < N; t++) wr[t] = 0xFFFFFFFFu;
            }
        } else {
            F::to_canonical(wr, F::mul(x, y));
        }
        break;
    case 16:
        if constexpr (C::REPR == REPR_U29_K256) {
            RowsK256 k;
            k.init();
            const auto x3 = F::add(F::dbl(x), x);
            const auto x7 = F::add(F::add(x3, x3), x);                  // limb magnitudes 3, 7
            const auto y2 = F::dbl(y);
            uint32_t a7 = 0, a3 = 0, b1 = 0, b2 = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {                               // row r takes the operands of the wave's lane r
                const uint32_t u7 = (uint32_t)__shfl((int)x7.e.v[t], (int)k.row, 64), u3 = (uint32_t)__shfl((int)x3.e.v[t], (int)k.row, 64);
                const uint32_t v1 = (uint32_t)__shfl((int)y.e.v[t], (int)k.row, 64), v2 = (uint32_t)__shfl((int)y2.e.v[t], (int)k.row, 64);
                a7 = k.pos == (uint32_t)t ? u7 : a7;
                a3 = k.pos == (uint32_t)t ? u3 : a3;
                b1 = k.pos == (uint32_t)t ? v1 : b1;
                b2 = k.pos == (uint32_t)t ? v2 : b2;
            }
            const uint32_t p1 = k.mul(a7, b1), p2 = k.mul(a3, b2);
            Fe<9> r1, r2;
            const uint32_t src = ((threadIdx.x & 3u) * 16u) * 4u;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                r1.v[t] = lane_pull(src + 4u * t, p1);
                r2.v[t] = lane_pull(src + 4u * t, p2);
            }
            F::to_canonical(wr, F::add(F::template wrap<1, 1>(r1), F::template wrap<1, 1>(r2)));
        } else {
            F::to_canonical(wr, F::mul(x, y));
        }
        break;
    case 11: {
        bool root;
        auto r = F::sqrt(x, &root);
        F::to_canonical(wr, root ? r : F::zero());
        break;
    }
    case 12: {
#pragma unroll 1
        for (int s = 0; s < 25; s++) {
            auto t = F::mul(x, y);
            auto u = F::norm(F::sub(F::add(t, x), F::dbl(y)));
            auto v = F::norm(F::neg(F::add(t, F::dbl(F::dbl(y)))));
            x = F::sqr(u);
            y = F::mul(v, F::one());
        }
        F::to_canonical(wr, F::add(x, y));
        break;
    }
    case 20:                                   // wire round trip: bytes -> words -> bytes
#pragma unroll
        for (int t = 0; t < N; t++) wr[t] = wa[t];
        break;
    case 21: F::to_canonical(wr, x); break;    // internal-domain round trip
    default:
#pragma unroll
        for (int t = 0; t < N; t++) wr[t] = 0;
        atomicOr(status, ST_BAD_SCALAR);
    }
    store_wire<C>(out + i * WB, wr);
}

// point: 0 P + Q (complete), 1 P + Q (complete, mixed), 2 2P, 3 -P, 4 P - Q, 5 P - Q (mixed), and the incomplete
// formulas of the ladders / the comb / the bucket sums on inputs inside their domain (P, Q finite, P != +-Q; the caller
// sees the identity otherwise): 6 2P by the Jacobian doubling, 7 P + Q by the Jacobian mixed addition, 8 P + Q by the
// XYZZ mixed addition, 9 P + Q by the XYZZ affine + affine addition; 10 (k256; n a multiple of 64) 32 P of the WAVE'S FIRST point in
// every lane, by five row-parallel complete doublings (ecgpu_rows.h RowsDblK256: what k_msm_combine's Horner chain runs on).
// Output: affine + identity flag (one inversion per lane).
template <class C>
__global__ void __launch_bounds__(BLOCK) k_selftest_point(int op, const uint8_t* __restrict__ pxy, const uint8_t* __restrict__ pinf,
                                                          const uint8_t* __restrict__ qxy, const uint8_t* __restrict__ qinf, size_t n,
                                                          uint8_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf, int* status) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe<C::NL> b = G::curve_b();
    Affine<C> pa, qa;
    const bool pf = load_affine<C>(&pa, pxy, pinf, i, b, status);
    const bool needs_q = op == 0 || op == 1 || op == 4 || op == 5 || op >= 7;
    const bool qf = needs_q && qxy ? load_affine<C>(&qa, qxy, qinf, i, b, status) : false;
    Proj<C> p = pf ? G::from_affine(pa) : G::identity(), r = G::identity();
    switch (op) {
    case 0: r = G::add(p, qf ? G::from_affine(qa) : G::identity(), b); break;
    case 1: r = qf ? G::add_mixed(p, qa, b) : p; break;
    case 2: r = G::dbl(p, b); break;
    case 3: r = G::neg(p); break;
    case 4: r = G::add(p, qf ? G::from_affine(qa) : G::identity(), b, true); break;
    case 5: r = qf ? G::add_mixed(p, qa, b, true) : p; break;
    case 6: if (pf) r = G::jac_to_proj(G::jac_dbl(G::jac_from_affine(pa))); break;
    case 7: if (pf && qf) r = G::jac_to_proj(G::jac_madd(G::jac_dbl(G::jac_from_affine(pa)), qa, false)); break;   // 2P + Q
    case 8: if (pf && qf) r = G::xyzz_to_proj(G::xyzz_madd(G::xyzz_from_affine(pa, false), qa, false)); break;
    case 9: if (pf && qf) r = G::xyzz_to_proj(G::xyzz_mmadd(pa, qa, false)); break;
    case 10:
        if constexpr (C::REPR == REPR_U29_K256) {
            __shared__ uint32_t rows_lds[BLOCK / 64][48];
            Fe<9> X, Y, Z;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                X.v[t] = (uint32_t)__shfl((int)p.x.v[t], 0, 64);
                Y.v[t] = (uint32_t)__shfl((int)p.y.v[t], 0, 64);
                Z.v[t] = (uint32_t)__shfl((int)p.z.v[t], 0, 64);
            }
            RowsDblK256 rd;
            rd.init();
            uint32_t A, B, Q = 0;
            rd.enter(rows_lds[threadIdx.x / 64], X, Y, Z, A, B);
#pragma unroll 1
            for (int s = 0; s < 5; s++) Q = rd.step(A, B);
            rd.leave(Q, X, Y, Z);
            r.x = X;
            r.y = F::norm(F::template wrap<2, 2>(Y)).e;
            r.z = Z;
        } else {
            atomicOr(status, ST_BAD_SCALAR);
        }
        break;
    default: atomicOr(status, ST_BAD_SCALAR);
    }
    if (G::is_identity(r)) {
        zero_wire<C>(out_xy + i * (2 * WB), 2);
        out_inf[i] = 1;
    } else {
        auto zi = F::inv(G::m(r.z));
        uint32_t w[N];
        F::to_canonical(w, F::mul(G::m(r.x), zi));
        store_wire<C>(out_xy + i * (2 * WB), w);
        F::to_canonical(w, F::mul(G::m(r.y), zi));
        store_wire<C>(out_xy + i * (2 * WB) + WB, w);
        out_inf[i] = 0;
    }
}

}  // namespace ecgpu
