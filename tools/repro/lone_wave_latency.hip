// lone_wave_latency.hip — what ONE wave alone on a SIMD pays per instruction on gfx950: the numbers behind the serial chains of the
// MSM tail (k_msm_combine's Horner chain, the running sums): dependent and independent v_mad_u64_u32 chains, the LDS crossbar
// (ds_bpermute_b32), DPP moves, and which lane a DPP row / wave shift reads from.
//     hipcc --offload-arch=gfx950 -O3 -o tools/repro/lone_wave_latency tools/repro/lone_wave_latency.hip && tools/repro/lone_wave_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int V>
__global__ void __launch_bounds__(64) k_chain(uint64_t* out, int iters, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t c0 = seed, c1 = seed + 1, c2 = seed + 2, c3 = seed + 3, c4 = seed + 4, c5 = seed + 5, c6 = seed + 6, c7 = seed + 7;
    uint32_t x = threadIdx.x, idx = ((threadIdx.x + 1) & 63) * 4;
    uint32_t y0 = x + 1, y1 = x + 2, y2 = x + 3, y3 = x + 4, y4 = x + 5, y5 = x + 6, y6 = x + 7;
    for (int i = 0; i < iters; i++) {
        if constexpr (V == 0) {            // 64 dependent multiply-adds
            REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (V == 1) {     // 64 multiply-adds on four accumulators in turn
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                                   "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");))
        } else if constexpr (V == 2) {     // ... on eight
            REP8(REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                                   "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                                   "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                                   "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");))
        } else if constexpr (V == 3) {     // 64 dependent trips through the LDS crossbar
            REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(idx));)
        } else if constexpr (V == 4) {     // 64 dependent DPP moves (row_shr:1)
            REP64(asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));)
        } else if constexpr (V == 5) {     // 64 dependent 32-bit additions
            REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));)
        } else if constexpr (V == 6) {     // 64 dependent 64-bit shift + add pairs (the carry step of a reduction)
            REP64(asm volatile("v_lshrrev_b64 %0, 29, %0\n\tv_lshl_add_u64 %0, %0, 0, %1" : "+v"(c0) : "v"(c1));)
        } else if constexpr (V == 8) {     // 64 dependent scalar additions (the scalar ALU: wave-uniform values)
            uint32_t sx = __builtin_amdgcn_readfirstlane(x), sa = __builtin_amdgcn_readfirstlane(a);
            REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(sx) : "s"(sa) : "scc");)
            x = sx;
        } else if constexpr (V == 9) {     // 64 scalar operations on four values in turn
            uint32_t s0 = __builtin_amdgcn_readfirstlane(x), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, sa = __builtin_amdgcn_readfirstlane(a);
            REP8(asm volatile("s_add_u32 %0, %0, %4\n\ts_xor_b32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_and_b32 %3, %3, %4\n\t"
                              "s_add_u32 %0, %0, %4\n\ts_xor_b32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_and_b32 %3, %3, %4"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(sa) : "scc");)
            x = s0 + s1 + s2 + s3;
        } else if constexpr (V == 10) {    // 64 32-bit vector additions on four values in turn
            REP8(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                              "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                              : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(a));)
        } else if constexpr (V == 7) {     // eight crossbar trips in flight, then one wait
            REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\t"
                              "ds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\t"
                              "s_waitcnt lgkmcnt(0)"
                              : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6) : "v"(idx));)
        }
    }
    out[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + x + a + b + y0 + y1 + y2 + y3 + y4 + y5 + y6;
}

__global__ void __launch_bounds__(64) k_dpp_lanes(uint32_t* out) {
    uint32_t lane = threadIdx.x, r;
    r = 999; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(lane)); out[lane] = r;
    r = 999; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(lane)); out[64 + lane] = r;
    r = 999; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r) : "v"(lane)); out[128 + lane] = r;
    r = 999; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(lane)); out[192 + lane] = r;
    r = 999; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r) : "v"(lane)); out[256 + lane] = r;
}

template <int V>
static void run(const char* what, int per_iter) {
    uint64_t* d;
    (void)hipMalloc(&d, 64 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k_chain<V>, dim3(1), dim3(64), 0, 0, d, 100, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chain<V>, dim3(1), dim3(64), 0, 0, d, iters, 7u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-62s %8.3f ms  %6.2f ns per instruction  = %5.1f cycles at 2.4 GHz\n", what, ms, ms * 1e6 / ((double)iters * per_iter),
           ms * 1e6 / ((double)iters * per_iter) * 2.4);
    (void)hipFree(d);
}

int main() {
    run<0>("v_mad_u64_u32, 64 dependent (one accumulator)", 64);
    run<1>("v_mad_u64_u32, four accumulators in turn", 256);
    run<2>("v_mad_u64_u32, eight accumulators in turn", 512);
    run<5>("v_add_u32, dependent", 64);
    run<10>("v_add_u32, four values in turn", 64);
    run<6>("v_lshrrev_b64 + v_lshl_add_u64, dependent pairs (per pair)", 64);
    run<8>("s_add_u32, dependent (scalar ALU)", 64);
    run<9>("s_add / s_xor / s_and, four values in turn (scalar ALU)", 64);
    run<3>("ds_bpermute_b32 + wait, dependent", 64);
    run<7>("ds_bpermute_b32, eight in flight per wait (per instruction)", 64);
    run<4>("v_mov_b32_dpp row_shr:1 (+ s_nop 1), dependent", 64);
    uint32_t* d;
    (void)hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k_dpp_lanes, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[320];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"row_shr:1", "row_shl:1", "row_shr:1 bound_ctrl:0", "wave_shr:1", "row_shr:2 bound_ctrl:0"};
    for (int v = 0; v < 5; v++) {
        printf("%-24s lane <- :", names[v]);
        for (int l = 0; l < 20; l++) printf(" %u", h[v * 64 + l]);
        printf(" ... (lanes 32..34:) %u %u %u\n", h[v * 64 + 32], h[v * 64 + 33], h[v * 64 + 34]);
    }
    return 0;
}
