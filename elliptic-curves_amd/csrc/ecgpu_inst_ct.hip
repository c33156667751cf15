// ecgpu_inst_ct.hip — instantiates the uniform-schedule kernels (ecgpu_ct.h) for -DECGPU_CURVE=...
#include "ecgpu_ct.h"
#include "ecgpu_launch.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

// ecgpu_lincomb_ct adds the products of k_var_base_ct with the tree kernel of the base group; instantiated here as well so
// that the translation unit tools/ct_isa_check.py compiles holds every kernel of the uniform-schedule entry points
template __global__ void k_proj_sum_level<CurveT>(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ out);

template <> int ct_base_luts<CurveT>() { return CT_BASE_LUTS<CurveT>; }
template <> void launch_var_base_ct<CurveT>(hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf, size_t n,
                                            uint32_t* tab, size_t slots, uint32_t* proj_out, uint8_t* flags, int* status) {
    hipLaunchKernelGGL(k_var_base_ct<CurveT>, dim3((unsigned)(slots / BLOCK)), dim3(BLOCK), 0, s, scalars, xy, inf, n, tab, slots,
                       proj_out, flags);
    hipLaunchKernelGGL(k_ct_flags, dim3(64), dim3(BLOCK), 0, s, (const uint8_t*)flags, n, status);
}
template <> void launch_fixed_base_ct<CurveT>(hipStream_t s, const uint8_t* scalars, size_t n, const uint32_t* lut, uint32_t* proj_out,
                                              uint8_t* flags, int* status) {
    hipLaunchKernelGGL(k_fixed_base_ct<CurveT>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, scalars, n, lut, proj_out,
                       flags);
    hipLaunchKernelGGL(k_ct_flags, dim3(64), dim3(BLOCK), 0, s, (const uint8_t*)flags, n, status);
}

}  // namespace ecgpu
