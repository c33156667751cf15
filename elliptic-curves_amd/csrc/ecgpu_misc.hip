// ecgpu_misc.hip — curve-independent kernels: k256 GLV split (parity probe) and VALU roof probes.
#include "ecgpu_kernels.h"
#include "ecgpu_launch.h"
#include "ecgpu_ecdsa.h"
#include "ecgpu_knobs.h"

#include <cstdlib>

namespace ecgpu {

// ecgpu_knobs.h: the environment is read by the tool build only
const char* knob(const char* name) {
#if defined(ECGPU_TUNING_KNOBS) && ECGPU_TUNING_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

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



// ---- exact per-instruction issue-rate probes (inline asm, 8 independent chains x 8 per loop trip) ----
// Reported as wave-instructions per second over the whole chip; divide (CUs x 4 SIMDs x clock) by it to get
// cycles per wave64 instruction per SIMD.
#define ISA_BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)
template <int WHICH>
__global__ void __launch_bounds__(BLOCK) k_isa_probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t a[8];
    uint32_t x[8];
    double f[8];
    uint32_t m = seed | 1u, m2 = (seed * 2654435761u) | 3u;
    double fm = 1.0000001, fc = 1e-9;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = ((uint64_t)(t + i) << 32) | (seed + i); x[i] = t * 7 + i; f[i] = (double)(t + i) * 1e-3; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
            if constexpr (WHICH == 0) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(m2) : "vcc");
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 1) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 2) {
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(x[i]), "+v"(x[(i + 4) & 7]) : "v"(m), "v"(m2) : "vcc");
                S(0) S(1) S(2) S(3)
#undef S
            } else if constexpr (WHICH == 3) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 4) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 5) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 6) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 7) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 8) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(m2));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 9) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 10) {
#define S(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(m2));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 11) {
#define S(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 12) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fc));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 13) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else if constexpr (WHICH == 14) {
#define S(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(m));
                ISA_BODY8(S)
#undef S
            } else {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(m) : );
                ISA_BODY8(S)
#undef S
            }
        }
    }
    uint64_t acc = 0;
    double facc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { acc ^= a[i] + x[i]; facc += f[i]; }
    if (acc == 0x1234567 || facc == 1.2345) out[t] = (uint32_t)acc;
}

void launch_isa_probe(hipStream_t s, int which, uint32_t* o, int blocks, int it) {
    switch (which) {
#define C(n) case n: hipLaunchKernelGGL(k_isa_probe<n>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14)
#undef C
    default: hipLaunchKernelGGL(k_isa_probe<15>, dim3(blocks), dim3(BLOCK), 0, s, o, it, 12345u); break;
    }
}

// ---- HBM gather probe -----------------------------------------------------------------------------------------
// The fixed-base kernel's memory access pattern in isolation: every lane reads `per_lane` pseudo-random 64-byte
// entries (4 x 16-byte vector loads, like load_packed_affine) of a table.  The byte count is known exactly
// (lanes x per_lane x 64), which calibrates rocprofv3's FETCH_SIZE for this pattern and measures the random-gather
// throughput the W = 24 comb table relies on.
__global__ void __launch_bounds__(BLOCK) k_gather_probe(const uint32_t* __restrict__ table, size_t entries, int per_lane,
                                                        uint32_t* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
    uint32_t acc = 0;
    for (int i = 0; i < per_lane; i++) {
        x ^= x >> 27; x *= 0x3C79AC492BA7B653ull; x ^= x >> 33;
        const uint32_t* e = table + (size_t)(x % entries) * 16;
        uint32_t w[16];
        load_words_vec<16>(w, e);
#pragma unroll
        for (int j = 0; j < 16; j++) acc ^= w[j];
    }
    out[t] = acc;
}
// The access pattern of the variable-base kernels' per-lane tables (ecgpu_var.h: [wave][entry][row][lane] u32, 8 entries x 30
// rows for the 10-limb curves), with KNOWN useful bytes: every wave writes its block once (240 row stores of 256 bytes) and
// then reads `reps` entries of 20 rows each — entry (r & 7) for all lanes (uniform: every load instruction is one
// contiguous 256-byte row) or a per-lane pseudo-random entry as a digit-dependent ladder does (scattered: the 64 lanes' words
// of one load instruction lie in up to 8 different rows).  Under rocprofv3 --pmc FETCH_SIZE the first calibrates the counter
// for this row width, the second measures how many bytes the memory system moves per useful byte when lanes pick different
// entries (tools/gpu_fetch_calibration.py).
__global__ void __launch_bounds__(BLOCK, 2) k_tabrow_probe(uint32_t* __restrict__ tab, int scattered, int reps, uint32_t* __restrict__ out) {
    constexpr int ROWS = 30, PER_ENTRY = ROWS * 64;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* base = tab + (t / 64) * (size_t)(8 * PER_ENTRY) + (t % 64);
    for (int e = 0; e < 8; e++)
        for (int r = 0; r < ROWS; r++) base[(size_t)e * PER_ENTRY + r * 64] = (uint32_t)(t * 2654435761u) + e * 31 + r;
    uint32_t acc = 0;
    uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
#pragma unroll 1
    for (int i = 0; i < reps; i++) {
        x ^= x >> 27; x *= 0x3C79AC492BA7B653ull; x ^= x >> 33;
        const int e = scattered ? (int)(x & 7) : (i & 7);
        const uint32_t* ent = base + (size_t)e * PER_ENTRY;
#pragma unroll
        for (int r = 0; r < 20; r++) acc ^= ent[r * 64];
    }
    out[t] = acc;
}
void launch_tabrow_probe(hipStream_t s, uint32_t* tab, int scattered, int reps, uint32_t* out, int blocks) {
    hipLaunchKernelGGL(k_tabrow_probe, dim3(blocks), dim3(BLOCK), 0, s, tab, scattered, reps, out);
}

void launch_gather_probe(hipStream_t s, const uint32_t* table, size_t entries, int per_lane, uint32_t* out, int blocks) {
    hipLaunchKernelGGL(k_gather_probe, dim3(blocks), dim3(BLOCK), 0, s, table, entries, per_lane, out);
}

void launch_schnorr_prepare_raw(hipStream_t s, const uint8_t* pk_x, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs,
                                size_t n, uint8_t* a, uint8_t* b, uint8_t* q_out, uint8_t* r_out, uint8_t* valid) {
    hipLaunchKernelGGL(k_schnorr_prepare_raw<K256Params>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, pk_x, msgs,
                       msg_len, sigs, n, a, b, q_out, r_out, valid);
}

void launch_sm2dsa_hash_msg(hipStream_t s, const uint8_t* distid, size_t distid_len, const uint8_t* q_xy, const uint8_t* msgs,
                            size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* e_out, uint8_t* r_out, uint8_t* s_out) {
    hipLaunchKernelGGL(k_sm2dsa_hash_msg<Sm2Params>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, distid, distid_len,
                       q_xy, msgs, msg_len, sigs, n, e_out, r_out, s_out);
}

void launch_bign_prepare(hipStream_t s, const uint8_t* h, const uint8_t* sigs, const uint8_t* q_xy, size_t n, uint8_t* a, uint8_t* b,
                         uint8_t* q_out, uint8_t* valid) {
    hipLaunchKernelGGL(k_bign_prepare<Bign256Params>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, h, sigs, q_xy, n, a,
                       b, q_out, valid);
}
void launch_bign_finish(hipStream_t s, const uint8_t* h, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* sigs,
                        const uint8_t* valid, size_t n, uint8_t* ok) {
    hipLaunchKernelGGL(k_bign_finish<Bign256Params>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, h, r_xy, r_inf, sigs,
                       valid, n, ok);
}
void launch_bign_hash_msg(hipStream_t s, const uint8_t* msgs, size_t msg_len, size_t n, uint8_t* h_out) {
    hipLaunchKernelGGL(k_bign_hash_msg, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, msgs, msg_len, n, h_out);
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
