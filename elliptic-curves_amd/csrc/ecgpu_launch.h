// ecgpu_launch.h — host-side launch wrappers, one explicit instantiation per curve.
// The kernels are large fully-inlined bodies; each (kernel group x curve) is its own translation
// unit (ecgpu_inst_*.hip compiled with -DECGPU_CURVE=...) so the build parallelises.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ecgpu_field.h"

namespace ecgpu {

struct MsmPlan {
    int c = 0, nwin = 0, seg = 0;
    size_t nb = 0;     // buckets per window = 2^(c-1)
    size_t nseg = 0;   // segments per window
    size_t ntiles = 0, tile = 0;   // counting-sort tiles (terms per tile)
    size_t chunk = 0, nchunks = 0; // accumulation: sorted entries per lane, lanes per window
    // optional timing marks of one launch (ecgpu_last_timing "prepare" / "finish" / "tree" / "combine"): recorded after k_msm_prepare,
    // after the bucket finish + running sums, after the tree over the segment sums; null = not recorded
    hipEvent_t detail[3] = {nullptr, nullptr, nullptr};
    size_t off_points = 0, off_digits = 0, off_vmask = 0, off_tilehist = 0, off_sorted = 0, off_count = 0, off_offset = 0,
           off_partials = 0, off_buckets = 0, off_segs = 0, off_wins = 0, off_biglist = 0;
    // two-level sort (large MSMs): level A partitions a window's entries by the top 8 bits of the bucket, level B sorts
    // by the remaining sort_bits_b bits inside the partitions.  sort_bits_b = 0: single-level sort.
    int sort_bits_b = 0;
    size_t npart = 0, ntiles2 = 0;
    // packed form of the two-level sort (round 4): ONE 32-bit word per entry between the levels — index | sign << idx_bits |
    // low bucket bits << (idx_bits + 1) — and level B as one workgroup per partition; needs idx_bits + 1 + sort_bits_b <= 32
    bool sort_packed = false;
    int idx_bits = 0;
    size_t off_tmpidx = 0, off_tmpkey = 0, off_count_a = 0, off_offset_a = 0, off_cursor = 0;
    size_t max_big = 0;            // upper bound on the number of buckets that have more than MSM_BIG_PARTIALS partial sums
    // sub-terms: MsmSplit<C>::SUB per term (k256: the two GLV halves), sub-term h of term i at index h * npad + i
    size_t npad = 0, nsub = 0;     // n rounded up to a multiple of 64; entries per window = SUB * npad
    int kbits = 0;                 // significant bits of a sub-scalar
    bool glv = false;              // k256: sub-terms are the GLV halves
    // per-window partial sums handed from launch_msm_parts to launch_msm_finish (and between GPUs): [nwin][nparts] points
    size_t nparts = 0, per_part = 0, off_parts = 0, parts_bytes = 0;
    size_t workspace_bytes = 0;
};

// ---- group "base": table construction, fixed-base kernel, normalisation, small helpers ----
template <class C> void launch_window_bases(hipStream_t s, uint32_t* bases, int w, int nwin);
template <class C> void launch_table_entries(hipStream_t s, const uint32_t* bases, uint32_t* entries, int w, int nwin);
// soa: proj / prefix quad-major (launch_fixed_base's soa form); wire output only
template <class C> void launch_normalize(hipStream_t s, bool out_internal, const uint32_t* proj, uint32_t* prefix, size_t n,
                                         uint8_t* out_xy, uint8_t* out_inf, uint32_t* out_limbs, bool soa = false);
template <class C> void launch_normalize_compressed(hipStream_t s, const uint32_t* proj, uint32_t* prefix, size_t n, uint8_t* out_x,
                                                    uint8_t* out_tag);
// soa: proj_out quad-major (store_proj_soa, ecgpu_kernels.h) for launch_normalize(..., soa = true); n must then be the record count
// the normalisation is launched with
template <class C> void launch_fixed_base(hipStream_t s, const uint8_t* scalars, size_t n, const uint32_t* table, int w,
                                          int nwin, uint32_t* proj_out, int* status, bool soa = false);
template <class C> void launch_load_proj(hipStream_t s, const uint8_t* xyz, size_t n, uint32_t* proj_out, int* status);
template <class C> void launch_point_sum(hipStream_t s, const uint8_t* xy, const uint8_t* inf, size_t n, uint32_t* proj_out,
                                         int* status);
// a[0] = sum of the n projective points a[0..n) (a is clobbered; tmp holds ceil(n / 256) points)
template <class C> void launch_proj_sum(hipStream_t s, uint32_t* a, size_t n, uint32_t* tmp);
// inv_prefix (n x N words) / inv_out (n wire scalars): scratch for the batch's inverses modulo the group order — s^-1 here, r^-1 in
// the recovery — by Montgomery's trick (k_scalar_batch_inv); both null: every lane inverts for itself
template <class C> void launch_ecdsa_prepare(hipStream_t s, const uint8_t* z, const uint8_t* r, const uint8_t* sig_s, const uint8_t* q_xy,
                                             size_t n, int reject_high_s, uint8_t* u1, uint8_t* u2, uint8_t* q_out, uint8_t* valid,
                                             uint32_t* inv_prefix = nullptr, uint8_t* inv_out = nullptr);
template <class C> void launch_ecdsa_hash_msg(hipStream_t s, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* z_out,
                                              uint8_t* r_out, uint8_t* s_out);
template <class C> void launch_ecdsa_recover_prepare(hipStream_t s, const uint8_t* z, const uint8_t* r, const uint8_t* sig_s,
                                                     const uint8_t* recid, size_t n, int reject_high_s, uint8_t* a, uint8_t* b,
                                                     uint8_t* q_out, uint8_t* valid, uint32_t* inv_prefix = nullptr,
                                                     uint8_t* inv_out = nullptr);
template <class C> void launch_ecdsa_recover_finish(hipStream_t s, uint8_t* xy, const uint8_t* inf, const uint8_t* valid, size_t n,
                                                    uint8_t* ok);
template <class C> void launch_sm2dsa_prepare(hipStream_t s, const uint8_t* r, const uint8_t* sig_s, const uint8_t* q_xy, size_t n, uint8_t* a,
                                              uint8_t* b, uint8_t* q_out, uint8_t* valid);
template <class C> void launch_sm2dsa_finish(hipStream_t s, const uint8_t* e, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* r,
                                             const uint8_t* valid, size_t n, uint8_t* ok);
template <class C> void launch_schnorr_prepare(hipStream_t s, const uint8_t* e, const uint8_t* r, const uint8_t* sig_s, const uint8_t* p_xy,
                                               size_t n, uint8_t* a, uint8_t* b, uint8_t* q_out, uint8_t* valid);
template <class C> void launch_schnorr_finish(hipStream_t s, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* r,
                                              const uint8_t* valid, size_t n, uint8_t* ok);
template <class C> void launch_extract_x(hipStream_t s, const uint8_t* xy, const uint8_t* inf, size_t n, uint8_t* out_x, uint8_t* ok);
template <class C> void launch_decompress(hipStream_t s, const uint8_t* xs, const uint8_t* y_is_odd, size_t n, uint8_t* out_xy,
                                          uint8_t* ok);
template <class C> void launch_decompress_tagged(hipStream_t s, const uint8_t* xs, const uint8_t* tags, size_t n, uint8_t* out_xy,
                                                 uint8_t* out_inf, int* status);
template <class C> void launch_ecdsa_finish(hipStream_t s, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* r,
                                            const uint8_t* valid, size_t n, uint8_t* ok);

template <class C> void launch_selftest_field(hipStream_t s, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, int* status);
template <class C> void launch_selftest_point(hipStream_t s, int op, const uint8_t* pxy, const uint8_t* pinf, const uint8_t* qxy,
                                              const uint8_t* qinf, size_t n, uint8_t* out_xy, uint8_t* out_inf, int* status);

// ---- group "var": variable-base kernel ----
template <class C> size_t var_base_slots(size_t n);     // table slots (threads) the launch will use
template <class C> size_t var_base_tab_words();         // 32-bit words of table scratch per slot
template <class C> void launch_var_base(hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf,
                                        size_t n, uint32_t* tab, size_t slots, uint32_t* proj_out, int* status,
                                        uint32_t* add_io = nullptr);      // add_io[i] += k[i] P[i] in place instead of proj_out[i] = k[i] P[i]

// ---- group "ct": uniform-schedule variants (ecgpu_ct.h); flags: n bytes of scratch (one verdict byte per element) ----
template <class C> int ct_base_luts();                  // generator LUTs: one per 6-bit window (CT_BASE_W), 32 affine entries each: lut i = {e * 2^(6 i) * G, e = 1..32}
template <class C> void launch_var_base_ct(hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf, size_t n,
                                           uint32_t* tab, size_t slots, uint32_t* proj_out, uint8_t* flags, int* status);
template <class C> void launch_fixed_base_ct(hipStream_t s, const uint8_t* scalars, size_t n, const uint32_t* lut, uint32_t* proj_out,
                                             uint8_t* flags, int* status);

// ---- group "msm": Pippenger pipeline ----
template <class C> MsmPlan msm_plan(size_t n, int force_c, bool glv);
template <class C> bool msm_use_glv(size_t n);          // k256: GLV halves for this term count?
template <class C> int msm_choose_window(size_t n);
template <class C> size_t msm_max_terms();              // sorted entries are sub-term index | sign << 31
template <class C> void launch_msm(const MsmPlan& p, hipStream_t s, const uint8_t* scalars, const uint8_t* xy,
                                   const uint8_t* inf, size_t n, void* workspace, uint32_t* out, int* status,
                                   hipEvent_t ev_sorted, hipEvent_t ev_accumulated, uint8_t* out_xy = nullptr, uint8_t* out_inf = nullptr);
template <class C> void launch_msm_parts(const MsmPlan& p, hipStream_t s, const uint8_t* scalars, const uint8_t* xy,
                                         const uint8_t* inf, size_t n, void* workspace, uint32_t* parts, int* status,
                                         hipEvent_t ev_sorted, hipEvent_t ev_accumulated);
// out_xy != nullptr: the result as a wire record (x || y, identity flag) straight from the last kernel; otherwise projective in out[0]
template <class C> void launch_msm_finish(const MsmPlan& p, hipStream_t s, const uint32_t* parts_all, int nranks, uint32_t* wins,
                                          uint32_t* out, uint8_t* out_xy = nullptr, uint8_t* out_inf = nullptr);

// ---- curve-independent ----
void launch_schnorr_prepare_raw(hipStream_t s, const uint8_t* pk_x, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs,
                                size_t n, uint8_t* a, uint8_t* b, uint8_t* q_out, uint8_t* r_out, uint8_t* valid);
// bign verification (bign-curve256v1 only; ecgpu_ecdsa.h)
void launch_bign_prepare(hipStream_t s, const uint8_t* h, const uint8_t* sigs, const uint8_t* q_xy, size_t n, uint8_t* a, uint8_t* b,
                         uint8_t* q_out, uint8_t* valid);
void launch_bign_finish(hipStream_t s, const uint8_t* h, const uint8_t* r_xy, const uint8_t* r_inf, const uint8_t* sigs,
                        const uint8_t* valid, size_t n, uint8_t* ok);
void launch_bign_hash_msg(hipStream_t s, const uint8_t* msgs, size_t msg_len, size_t n, uint8_t* h_out);
void launch_sm2dsa_hash_msg(hipStream_t s, const uint8_t* distid, size_t distid_len, const uint8_t* q_xy, const uint8_t* msgs,
                            size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* e_out, uint8_t* r_out, uint8_t* s_out);
void launch_k256_glv(hipStream_t s, const uint8_t* scalars, size_t n, uint8_t* r1, uint8_t* r2, int* status);
void launch_valu_probe(hipStream_t s, int which, uint32_t* out, int blocks, int iters);
void launch_isa_probe(hipStream_t s, int which, uint32_t* out, int blocks, int iters);
void launch_gather_probe(hipStream_t s, const uint32_t* table, size_t entries, int per_lane, uint32_t* out, int blocks);
void launch_tabrow_probe(hipStream_t s, uint32_t* tab, int scattered, int reps, uint32_t* out, int blocks);

}  // namespace ecgpu
