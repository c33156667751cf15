// ecgpu_var.h — variable-base batch scalar multiplication kernel (HIP only).
#pragma once

#include "ecgpu_kernels.h"
#include "ecgpu_varmul.h"

namespace ecgpu {

// ---- variable base: out[i] = k[i] * P[i] -------------------------------------------------------------
// One lane per scalar multiplication (ecgpu_varmul.h has the algorithm and the reference citations).  The
// 8-entry table lives in HBM scratch instead of the CPU stack.
constexpr int VAR_TAB_ELEMS = 3;   // x, y of the affine entry + the Z ratio while the table is being built

template <class C>
struct VarTabHbm {
    // [wave][entry][row][lane]: one wave's rows are 256 contiguous bytes, and every row of an entry is a
    // compile-time offset from the entry's base, so a whole entry costs one address computation
    uint32_t* base;     // wave block + lane
    static constexpr int NL = C::NL;
    static constexpr int ROWS = VAR_TAB_ELEMS * NL;
    __device__ void put_el(int e, int k, const Fe<NL>& v) {
        uint32_t* row = base + (size_t)e * (ROWS * 64) + (size_t)k * (NL * 64);
#pragma unroll
        for (int l = 0; l < NL; l++) row[l * 64] = v.v[l];
    }
    __device__ Fe<NL> get_el(int e, int k) const {
        const uint32_t* row = base + (size_t)e * (ROWS * 64) + (size_t)k * (NL * 64);
        Fe<NL> v;
#pragma unroll
        for (int l = 0; l < NL; l++) v.v[l] = row[l * 64];
        return v;
    }
};

// ADD: the verification shape (add_io below) as an instantiation of its own — the plain kernel's register allocation is not to
// know about it (with the branch inside one kernel p384's spilled registers went from 146 to 275 and the ladder slowed by 1.5 %)
template <class C, bool ADD = false>
__global__ void __launch_bounds__(BLOCK, 2)
k_var_base(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points_xy,
           const uint8_t* __restrict__ points_inf, size_t n, uint32_t* __restrict__ tab, size_t tstride,
           uint32_t* __restrict__ proj_out, int* status, uint32_t* __restrict__ add_io) {
    using G = Group<C>;
    constexpr int N = C::N, NL = C::NL;
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // < tstride
    Fe<NL> b = G::curve_b();
    VarTabHbm<C> io{tab + (slot / 64) * (size_t)(8 * VarTabHbm<C>::ROWS * 64) + (slot % 64)};
    for (size_t i = slot; i < n; i += tstride) {
        uint32_t k[N];
        load_scalar<C>(k, scalars, i, status);
        Affine<C> a;
        bool finite = load_affine<C>(&a, points_xy, points_inf, i, b, status);
        // add_io: the verification shape a G + b P (`mul_by_generator_and_mul_add_vartime`, primeorder/src/mul_backend.rs:29-40,
        // k256/src/arithmetic/mul.rs:303-310 — one loop in the reference): a G is in add_io[i] already (k_fixed_base), the product is
        // added to it in place — no launch of its own for the addition, no second projective array through HBM
        if constexpr (ADD) {
            if (finite) {                       // (the product first: a G is loaded when the ladder's registers are free again)
                const Proj<C> kp = var_base_mul<C>(a, k, b, io);
                store_proj<C>(add_io, i, G::add(load_proj<C>(add_io, i), kp, b));
            }
            continue;
        }
        if (!finite) {
            store_proj<C>(proj_out, i, G::identity());
            continue;
        }
        store_proj<C>(proj_out, i, var_base_mul<C>(a, k, b, io));
    }
}

}  // namespace ecgpu
