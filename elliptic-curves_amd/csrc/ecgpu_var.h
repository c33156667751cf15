// ecgpu_var.h — variable-base batch scalar multiplication kernel (HIP only).
#pragma once

#include "ecgpu_kernels.h"

namespace ecgpu {

// ---- variable base: out[i] = k[i] * P[i] -------------------------------------------------------------
// Drop-in for `ProjectivePoint * Scalar` (primeorder projective.rs:133-137 + lincomb :532-557; k256
// mul.rs:236-238).  Same structure as the reference: per-point table [P..8P], signed radix-16 digits
// (bit-identical to Radix16Decomposition via Radix16Msb), 4 doublings + 1 table addition per digit.
// The 8-entry table lives in HBM scratch, lane-minor, instead of the CPU stack; a negative digit is
// folded into the addition formula.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_var_base(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points_xy,
           const uint8_t* __restrict__ points_inf, size_t n, uint32_t* __restrict__ tab, size_t tstride,
           uint32_t* __restrict__ proj_out, int* status) {
    using G = Group<C>;
    constexpr int N = C::N, NL = C::NL;
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // < tstride
    Fe<NL> b = G::curve_b();
    for (size_t i = slot; i < n; i += tstride) {
        uint32_t k[N];
        load_scalar<C>(k, scalars, i, status);
        Affine<C> a;
        bool finite = load_affine<C>(&a, points_xy, points_inf, i, b, status);
        if (!finite) {
            store_proj<C>(proj_out, i, G::identity());
            continue;
        }
        // table: e*P for e = 1..8  (LookupTable::new, primeorder/src/tables/lookup.rs:30-38)
        Proj<C> m = G::from_affine(a);
#pragma unroll 1
        for (int e = 0; e < 8; e++) {
            uint32_t* row = tab + (size_t)e * (3 * NL) * tstride + slot;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                row[(size_t)l * tstride] = m.x.v[l];
                row[(size_t)(NL + l) * tstride] = m.y.v[l];
                row[(size_t)(2 * NL + l) * tstride] = m.z.v[l];
            }
            if (e < 7) m = G::add_mixed(m, a, b);
        }
        Radix16Msb<N> digits;
        digits.init(k);
        Proj<C> acc = G::identity();
#pragma unroll 1
        for (int di = 8 * N; di >= 0; di--) {
            if (di != 8 * N) {
                acc = G::dbl(acc, b); acc = G::dbl(acc, b);
                acc = G::dbl(acc, b); acc = G::dbl(acc, b);
            }
            int d = digits.digit(di);
            if (d != 0) {
                uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                const uint32_t* row = tab + (size_t)(mag - 1) * (3 * NL) * tstride + slot;
                Proj<C> q;
#pragma unroll
                for (int l = 0; l < NL; l++) {
                    q.x.v[l] = row[(size_t)l * tstride];
                    q.y.v[l] = row[(size_t)(NL + l) * tstride];
                    q.z.v[l] = row[(size_t)(2 * NL + l) * tstride];
                }
                acc = G::add(acc, q, b, d < 0);
            }
        }
        store_proj<C>(proj_out, i, acc);
    }
}

}  // namespace ecgpu
