// p521_byte_loads.hip — stand-alone probe for the wire-record load fault of docs/DESIGN_long_form_r01-r05.md §4 (round 2).
//
// History: p521's 66-byte big-endian records were first decoded byte by byte (`words[j] |= bytes[k] << s`).  hipcc merges such
// loads into wide unaligned ones (8 + 2 + 16 + 16 + 16 + 8 bytes) and re-extracts the bytes with v_perm_b32 / SDWA sequences;
// inside two large kernels (k_selftest_field<P521Params>, then k_ecdsa_recover_prepare<P521Params>) the decoded operand
// came out with wrong bits on gfx950 while the host build of the same source was right
// (profiles/r02/diag_recover_p521*.txt).  The library now reads one halfword + sixteen whole words (ecgpu_kernels.h
// load_wire) and tools/wire_codec_isa_check.py verifies that shape on the ISA.
//
// This file is the reduction attempt: ONLY the old byte-wise decoder, in three kernel shapes (plain copy; decode followed by
// enough dependent arithmetic to raise the register pressure; two records per lane with the words kept live across a loop),
// each checked word for word against the host.  Build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/p521_byte_loads tools/repro/p521_byte_loads.hip && /tmp/p521_byte_loads
// Exit status = number of shapes that decoded a record wrongly (0: the fault needs more context than these shapes).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int WB = 66, NW = 17;

// the decoder as it was at the time of the fault: byte by byte, most significant byte first
__host__ __device__ inline void load_be_bytes(uint32_t* words, const uint8_t* bytes) {
#pragma unroll
    for (int j = 0; j < NW; j++) words[j] = 0;
#pragma unroll
    for (int k = 0; k < WB; k++) {
        const int bit = 8 * (WB - 1 - k);
        words[bit / 32] |= (uint32_t)bytes[k] << (bit % 32);
    }
}

__global__ void k_copy(const uint8_t* in, size_t n, uint32_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[NW];
    load_be_bytes(w, in + i * WB);
    for (int j = 0; j < NW; j++) out[i * NW + j] = w[j];
}

// decode, then a chain of multiply-adds over all words (many live registers), then undo it: out must equal the record
__global__ void k_pressure(const uint8_t* in, size_t n, uint32_t* out, int rounds) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[NW], acc[NW];
    load_be_bytes(w, in + i * WB);
    for (int j = 0; j < NW; j++) acc[j] = w[j];
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const uint64_t t = (uint64_t)acc[j] * (uint32_t)(2 * r + 3) + acc[(j + 5) % NW];
            acc[j] = (uint32_t)t ^ (uint32_t)(t >> 32);
        }
    }
    uint32_t mix = 0;
    for (int j = 0; j < NW; j++) mix ^= acc[j];
    for (int j = 0; j < NW; j++) out[i * NW + j] = w[j] + (mix & 0u);        // the chain stays live, the result is w
}

// two records per lane (a value and a modulus-like operand), compared and conditionally subtracted like a range check
__global__ void k_two(const uint8_t* a, const uint8_t* b, size_t n, uint32_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[NW], y[NW];
    load_be_bytes(x, a + i * WB);
    load_be_bytes(y, b + i * WB);
    uint32_t borrow = 0, d[NW];
    for (int j = 0; j < NW; j++) {
        const uint64_t t = (uint64_t)x[j] - y[j] - borrow;
        d[j] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    for (int j = 0; j < NW; j++) out[i * NW + j] = borrow ? x[j] : d[j];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(r_)); return 99; } } while (0)

int main() {
    const size_t n = 1 << 16;
    std::vector<uint8_t> a(n * WB), b(n * WB);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint8_t)(s >> 32); };
    for (auto& v : a) v = next();
    for (auto& v : b) v = next();
    for (size_t i = 0; i < n; i++) { a[i * WB] &= 1; b[i * WB] &= 1; }          // 521-bit values: the top byte holds one bit
    uint8_t *da, *db;
    uint32_t* dout;
    CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dout, n * NW * 4));
    CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
    std::vector<uint32_t> got(n * NW);
    int bad_shapes = 0;
    for (int shape = 0; shape < 3; shape++) {
        CK(hipMemset(dout, 0xEE, n * NW * 4));
        if (shape == 0) hipLaunchKernelGGL(k_copy, dim3(n / 256), dim3(256), 0, 0, da, n, dout);
        if (shape == 1) hipLaunchKernelGGL(k_pressure, dim3(n / 256), dim3(256), 0, 0, da, n, dout, 40);
        if (shape == 2) hipLaunchKernelGGL(k_two, dim3(n / 256), dim3(256), 0, 0, da, db, n, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, n * NW * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t x[NW], y[NW], want[NW];
            load_be_bytes(x, a.data() + i * WB);
            if (shape == 2) {
                load_be_bytes(y, b.data() + i * WB);
                uint32_t borrow = 0, d[NW];
                for (int j = 0; j < NW; j++) { const uint64_t t = (uint64_t)x[j] - y[j] - borrow; d[j] = (uint32_t)t; borrow = (uint32_t)(t >> 63); }
                for (int j = 0; j < NW; j++) want[j] = borrow ? x[j] : d[j];
            } else {
                for (int j = 0; j < NW; j++) want[j] = x[j];
            }
            bool ok = true;
            for (int j = 0; j < NW; j++) ok = ok && got[i * NW + j] == want[j];
            bad += ok ? 0 : 1;
        }
        std::printf("shape %d (%s): %zu of %zu records decoded wrongly\n", shape, shape == 0 ? "copy" : shape == 1 ? "register pressure" : "two records + range check", bad, n);
        bad_shapes += bad ? 1 : 0;
    }
    return bad_shapes;
}
