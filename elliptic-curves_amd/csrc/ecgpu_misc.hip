// ecgpu_misc.hip — curve-independent kernels: k256 GLV split (parity probe) and VALU roof probes.
#include "ecgpu_kernels.h"
#include "ecgpu_launch.h"

namespace ecgpu {

// k256 GLV split, exposed for parity checks against glv::decompose_scalar
__global__ void k_k256_glv(const uint8_t* scalars, size_t n, uint8_t* r1_out, uint8_t* r2_out, int* status) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8], r1[8], r2[8];
    load_scalar<K256Params>(k, scalars, i, status);
    K256Scalar::decompose(r1, r2, k);
    store_be_vec<8>(r1_out + i * 32, r1);
    store_be_vec<8>(r2_out + i * 32, r2);
}

// ---- integer-VALU roof probes ---------------------------------------------------------------------------
// Dependency-light instruction streams (8 independent chains per lane) used to pin the peak rate of
// the multiply/add instructions the field arithmetic is made of.
template <int WHICH>
__global__ void __launch_bounds__(BLOCK) k_valu_probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t a[8];
    uint32_t m = seed | 1u, m2 = (seed * 2654435761u) | 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = ((uint64_t)(t + i) << 32) | (seed + i);
    double fa[8];
#pragma unroll
    for (int i = 0; i < 8; i++) fa[i] = (double)(t + i) * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (WHICH == 0) {
                    a[i] = (uint64_t)(uint32_t)a[i] * m + a[i];                          // v_mad_u64_u32
                } else if constexpr (WHICH == 1) {
                    a[i] = (uint32_t)a[i] * m2;                                           // v_mul_lo_u32
                } else if constexpr (WHICH == 2) {
                    a[i] = __umulhi((uint32_t)a[i], m2) + 1u;                             // v_mul_hi_u32 (+add)
                } else if constexpr (WHICH == 3) {
                    a[i] = (uint32_t)a[i] + m;                                            // v_add_u32
                } else if constexpr (WHICH == 4) {
                    a[i] = a[i] + ((uint64_t)m << 7 | m2);                                // 64-bit add
                } else if constexpr (WHICH == 5) {
                    a[i] = __umul24((uint32_t)a[i], m) + m2;                              // v_mad_u32_u24
                } else if constexpr (WHICH == 6) {
                    fa[i] = __builtin_fma(fa[i], 1.0000001, 1e-9);                        // v_fma_f64
                } else {
                    a[i] = a[i] + (a[(i + 1) & 7] >> 1);                                  // add/shift mix
                }
            }
        }
    }
    uint64_t acc = 0;
    double facc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { acc ^= a[i]; facc += fa[i]; }
    if (acc == 0x1234567 || facc == 1.2345) out[t] = (uint32_t)acc;  // keep the chains alive
}


void launch_k256_glv(hipStream_t s, const uint8_t* scalars, size_t n, uint8_t* r1, uint8_t* r2, int* status) {
    hipLaunchKernelGGL(k_k256_glv, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, scalars, n, r1, r2, status);
}

void launch_valu_probe(hipStream_t s, int which, uint32_t* o, int blocks, int it) {
    switch (which) {
    case 0: hipLaunchKernelGGL(k_valu_probe<0>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 1: hipLaunchKernelGGL(k_valu_probe<1>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 2: hipLaunchKernelGGL(k_valu_probe<2>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 3: hipLaunchKernelGGL(k_valu_probe<3>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 4: hipLaunchKernelGGL(k_valu_probe<4>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 5: hipLaunchKernelGGL(k_valu_probe<5>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    case 6: hipLaunchKernelGGL(k_valu_probe<6>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    default: hipLaunchKernelGGL(k_valu_probe<7>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    }
}

}  // namespace ecgpu
