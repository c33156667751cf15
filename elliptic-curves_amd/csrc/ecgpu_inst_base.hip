// ecgpu_inst_base.hip — instantiates the "base" kernel group for -DECGPU_CURVE=<K256Params|P256Params|P384Params>.
#include <cstdlib>

#include "ecgpu_kernels.h"
#include "ecgpu_ecdsa.h"
#include "ecgpu_launch.h"
#include "ecgpu_knobs.h"
#include "ecgpu_selftest.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

static inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

template <> void launch_window_bases<CurveT>(hipStream_t s, uint32_t* bases, int w, int nwin) {
    hipLaunchKernelGGL(k_window_bases<CurveT>, dim3(1), dim3(64), 0, s, bases, w, nwin);
}
template <> void launch_table_entries<CurveT>(hipStream_t s, const uint32_t* bases, uint32_t* entries, int w, int nwin) {
    int tlog = w - 1 > 6 ? w - 1 - 6 : 0;                     // 64 entries per lane ...
    if (tlog > 17) tlog = 17;                                 // ... but no more lanes than fill the machine a few times
    size_t total = ((size_t)1 << tlog) * nwin;
    hipLaunchKernelGGL(k_table_entries<CurveT>, dim3(grid_for(total)), dim3(BLOCK), 0, s, bases, entries, w, nwin, tlog);
}
// One inversion per lane, amortised over K points (Montgomery's trick).  The inversion is a serial chain of ~45k
// instructions whatever K is, so the kernel is fastest when there is about one wave per SIMD (1024 of them):
// K = ceil(n / 65536), capped at 64 for large batches.
template <> void launch_normalize<CurveT>(hipStream_t s, bool out_internal, const uint32_t* proj, uint32_t* prefix, size_t n,
                                          uint8_t* out_xy, uint8_t* out_inf, uint32_t* out_limbs, bool soa) {
    if (n == 0) return;
    size_t k = (n + 65535) / 65536;
    if (k > 64) k = 64;
    if (const char* e = knob("ECGPU_NORM_K")) {          // tuning knob: points per lane
        long v = atol(e);
        if (v >= 1 && v <= 1024) k = (size_t)v;
    }
    size_t nthreads = (n + k - 1) / k;
    if (soa && !out_internal)
        hipLaunchKernelGGL((k_normalize<CurveT, NORM_WIRE, true>), dim3(grid_for(nthreads)), dim3(BLOCK), 0, s, proj, prefix, n, nthreads,
                           out_xy, out_inf, out_limbs);
    else if (out_internal)
        hipLaunchKernelGGL((k_normalize<CurveT, NORM_PACKED>), dim3(grid_for(nthreads)), dim3(BLOCK), 0, s, proj, prefix, n, nthreads,
                           out_xy, out_inf, out_limbs);
    else
        hipLaunchKernelGGL((k_normalize<CurveT, NORM_WIRE>), dim3(grid_for(nthreads)), dim3(BLOCK), 0, s, proj, prefix, n, nthreads,
                           out_xy, out_inf, out_limbs);
}
template <> void launch_normalize_compressed<CurveT>(hipStream_t s, const uint32_t* proj, uint32_t* prefix, size_t n,
                                                     uint8_t* out_x, uint8_t* out_tag) {
    if (n == 0) return;
    size_t k = (n + 65535) / 65536;
    if (k > 64) k = 64;
    size_t nthreads = (n + k - 1) / k;
    hipLaunchKernelGGL((k_normalize<CurveT, NORM_COMPRESSED>), dim3(grid_for(nthreads)), dim3(BLOCK), 0, s, proj, prefix, n, nthreads,
                       out_x, out_tag, (uint32_t*)nullptr);
}
template <> void launch_fixed_base<CurveT>(hipStream_t s, const uint8_t* scalars, size_t n, const uint32_t* table, int w,
                                           int nwin, uint32_t* proj_out, int* status, bool soa) {
    if (soa)
        hipLaunchKernelGGL((k_fixed_base<CurveT, true>), dim3(grid_for(n)), dim3(BLOCK), 0, s, scalars, n, table, w, nwin, proj_out, status);
    else
        hipLaunchKernelGGL((k_fixed_base<CurveT, false>), dim3(grid_for(n)), dim3(BLOCK), 0, s, scalars, n, table, w, nwin, proj_out, status);
}
template <> void launch_load_proj<CurveT>(hipStream_t s, const uint8_t* xyz, size_t n, uint32_t* proj_out, int* status) {
    hipLaunchKernelGGL(k_load_proj<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, xyz, n, proj_out, status);
}
template <> void launch_point_sum<CurveT>(hipStream_t s, const uint8_t* xy, const uint8_t* inf, size_t n, uint32_t* proj_out,
                                          int* status) {
    hipLaunchKernelGGL(k_point_sum<CurveT>, dim3(1), dim3(BLOCK), 0, s, xy, inf, n, proj_out, status);
}
template <> void launch_proj_sum<CurveT>(hipStream_t s, uint32_t* a, size_t n, uint32_t* tmp) {
    uint32_t *src = a, *dst = tmp;
    size_t m = n;
    while (m > 1) {
        const size_t g = (m + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(k_proj_sum_level<CurveT>, dim3((unsigned)g), dim3(BLOCK), 0, s, (const uint32_t*)src, m, dst);
        m = g;
        uint32_t* t = src; src = dst; dst = t;
    }
    if (src != a) (void)hipMemcpyAsync(a, src, 3 * Field<CurveT>::NS * 4, hipMemcpyDeviceToDevice, s);
}
// the batch's inverses modulo the group order: like the normalisation, about one wave per SIMD (one inversion per lane)
static void launch_scalar_batch_inv(hipStream_t s, const uint8_t* in, size_t n, uint32_t* prefix, uint8_t* out) {
    if (n == 0) return;
    size_t k = (n + 65535) / 65536;
    if (k > 64) k = 64;
    const size_t nthreads = (n + k - 1) / k;
    hipLaunchKernelGGL(k_scalar_batch_inv<CurveT>, dim3(grid_for(nthreads)), dim3(BLOCK), 0, s, in, n, nthreads, prefix, out);
}
template <> void launch_ecdsa_prepare<CurveT>(hipStream_t s, const uint8_t* z, const uint8_t* r, const uint8_t* sig_s,
                                              const uint8_t* q_xy, size_t n, int reject_high_s, uint8_t* u1, uint8_t* u2,
                                              uint8_t* q_out, uint8_t* valid, uint32_t* inv_prefix, uint8_t* inv_out) {
    const bool batch = inv_prefix && inv_out;
    if (batch) launch_scalar_batch_inv(s, sig_s, n, inv_prefix, inv_out);
    hipLaunchKernelGGL(k_ecdsa_prepare<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, z, r, sig_s, q_xy, n, reject_high_s, u1, u2,
                       q_out, valid, batch ? (const uint8_t*)inv_out : (const uint8_t*)nullptr);
}
template <> void launch_schnorr_prepare<CurveT>(hipStream_t s, const uint8_t* e, const uint8_t* r, const uint8_t* sig_s,
                                                const uint8_t* p_xy, size_t n, uint8_t* a, uint8_t* b, uint8_t* q_out,
                                                uint8_t* valid) {
    hipLaunchKernelGGL(k_schnorr_prepare<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, e, r, sig_s, p_xy, n, a, b, q_out, valid);
}
template <> void launch_schnorr_finish<CurveT>(hipStream_t s, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* r,
                                               const uint8_t* valid, size_t n, uint8_t* ok) {
    hipLaunchKernelGGL(k_schnorr_finish<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, r_xy, r_inf, r, valid, n, ok);
}
template <> void launch_extract_x<CurveT>(hipStream_t s, const uint8_t* xy, const uint8_t* inf, size_t n, uint8_t* out_x,
                                          uint8_t* ok) {
    hipLaunchKernelGGL(k_extract_x<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, xy, inf, n, out_x, ok);
}
template <> void launch_decompress<CurveT>(hipStream_t s, const uint8_t* xs, const uint8_t* y_is_odd, size_t n,
                                           uint8_t* out_xy, uint8_t* ok) {
    hipLaunchKernelGGL(k_decompress<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, xs, y_is_odd, n, out_xy, ok);
}
template <> void launch_decompress_tagged<CurveT>(hipStream_t s, const uint8_t* xs, const uint8_t* tags, size_t n, uint8_t* out_xy,
                                                  uint8_t* out_inf, int* status) {
    hipLaunchKernelGGL(k_decompress_tagged<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, xs, tags, n, out_xy, out_inf, status);
}
template <> void launch_ecdsa_finish<CurveT>(hipStream_t s, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* r,
                                             const uint8_t* valid, size_t n, uint8_t* ok) {
    hipLaunchKernelGGL(k_ecdsa_finish<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, r_xy, r_inf, r, valid, n, ok);
}

template <> void launch_ecdsa_hash_msg<CurveT>(hipStream_t s, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs, size_t n,
                                               uint8_t* z_out, uint8_t* r_out, uint8_t* s_out) {
    hipLaunchKernelGGL(k_ecdsa_hash_msg<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, msgs, msg_len, sigs, n, z_out, r_out, s_out);
}
template <> void launch_ecdsa_recover_prepare<CurveT>(hipStream_t s, const uint8_t* z, const uint8_t* r, const uint8_t* sig_s,
                                                      const uint8_t* recid, size_t n, int reject_high_s, uint8_t* a, uint8_t* b,
                                                      uint8_t* q_out, uint8_t* valid, uint32_t* inv_prefix, uint8_t* inv_out) {
    const bool batch = inv_prefix && inv_out;
    if (batch) launch_scalar_batch_inv(s, r, n, inv_prefix, inv_out);
    hipLaunchKernelGGL(k_ecdsa_recover_prepare<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, z, r, sig_s, recid, n, reject_high_s,
                       a, b, q_out, valid, batch ? (const uint8_t*)inv_out : (const uint8_t*)nullptr);
}
template <> void launch_ecdsa_recover_finish<CurveT>(hipStream_t s, uint8_t* xy, const uint8_t* inf, const uint8_t* valid, size_t n,
                                                     uint8_t* ok) {
    hipLaunchKernelGGL(k_ecdsa_recover_finish<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, xy, inf, valid, n, ok);
}
template <> void launch_sm2dsa_prepare<CurveT>(hipStream_t s, const uint8_t* r, const uint8_t* sig_s, const uint8_t* q_xy, size_t n,
                                               uint8_t* a, uint8_t* b, uint8_t* q_out, uint8_t* valid) {
    hipLaunchKernelGGL(k_sm2dsa_prepare<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, r, sig_s, q_xy, n, a, b, q_out, valid);
}
template <> void launch_sm2dsa_finish<CurveT>(hipStream_t s, const uint8_t* e, const uint8_t* r_xy, const uint8_t* r_inf,
                                              const uint8_t* r, const uint8_t* valid, size_t n, uint8_t* ok) {
    hipLaunchKernelGGL(k_sm2dsa_finish<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, e, r_xy, r_inf, r, valid, n, ok);
}
template <> void launch_selftest_field<CurveT>(hipStream_t s, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out,
                                               int* status) {
    hipLaunchKernelGGL(k_selftest_field<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, op, a, b, n, out, status);
}
template <> void launch_selftest_point<CurveT>(hipStream_t s, int op, const uint8_t* pxy, const uint8_t* pinf, const uint8_t* qxy,
                                               const uint8_t* qinf, size_t n, uint8_t* out_xy, uint8_t* out_inf, int* status) {
    hipLaunchKernelGGL(k_selftest_point<CurveT>, dim3(grid_for(n)), dim3(BLOCK), 0, s, op, pxy, pinf, qxy, qinf, n, out_xy, out_inf,
                       status);
}

}  // namespace ecgpu
