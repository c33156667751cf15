// ecgpu_msm.h — Pippenger multi-scalar multiplication for gfx950 (HIP only).
//
// Replaces `LinearCombination::lincomb` / `lincomb_vartime` (k256/src/arithmetic/mul.rs:84-163,
// primeorder/src/projective.rs:480-557, wnaf/src/lib.rs:157-194).  The reference is Straus
// interleaving: Theta(N * bits/5) additions and ~2 KB of tables per term.  On the GPU the same
// group element is computed with the bucket method:
//
//   sum_i k_i P_i = sum_w 2^(c w) * sum_b weight(b) * ( sum_{i : bucket_w(k_i) = b} +-P_i )
//
//   prepare     one lane per term: decode + validate scalar and point once, keep the point in packed
//               internal form, recode k into nwin 16-bit (bucket, sign) digits (MsmDigitStream = msm_digit's digits in window order,
//               ecgpu_recode.h; k256 up to 2^21 terms: on the two GLV halves, K256Scalar::decompose_signed)
//   sort        counting sort of (sign, term index) by bucket, per window; every bucket becomes a contiguous run, so
//               accumulation needs no atomics and no conflict handling.  Below 2^17 entries per window in ONE level, with
//               the histogram of a whole window (2^(c-1) counters = 128 KiB at c = 16) held in one workgroup's LDS: a
//               workgroup owns a (tile of terms, window) pair, counts with LDS atomics (hist), a scan over tiles and
//               buckets turns the counts into offsets (tile_scan, scan), and the same workgroup shape scatters with LDS
//               cursors (scatter).  From 2^17 entries on in TWO levels (k_msm_sort_a / k_msm_sort_b below: partition by the
//               top bucket bits with LDS-staged coalesced run writes and one packed 32-bit word per entry, then one
//               workgroup per partition)
//   accumulate  one lane per `chunk` consecutive sorted entries of a window (ecgpu_msm_chunk.h): mixed XYZZ additions with an
//               exactness test per stretch, a partial sum written at every bucket boundary   <- the hot loop; perfectly balanced
//               for ANY scalar distribution
//   finish      one lane per (window, bucket): adds the bucket's 1-2 partial sums; a bucket with more than 32 of them
//               (degenerate inputs) is handed to a whole workgroup
//   reduce      running-sum trick on segments of buckets, segment sums, window sums (LDS trees; k256: additions on quad lanes)
//   combine     Horner over the windows (c doublings each) on ONE wave: k256 complete doublings / additions on quad lanes in
//               homogeneous coordinates (msm_hom_dbl_quad / msm_hom_add_quad), the a = -3 sets Jacobian doublings on three
//               lanes; the result leaves as an affine wire record
//
// Workspace layout (one allocation, offsets in MsmPlan): packed affine points [n][2N] u32,
// digits [nwin][n] u16 + validity bits [nwin][n/64] u64, tile histograms [nwin][ntiles][NB] u32 (one-level sort) / the level-A
// output [nwin][n] u32 + partition counts and offsets (two-level sort), sorted [nwin][n] u32, counts/offsets [nwin][NB]
// u32, partial sums [nwin][NB + nchunks][3 NS], buckets [nwin][NB][3 NS], segment sums, window sums.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "ecgpu_kernels.h"
#include "ecgpu_launch.h"
#include "ecgpu_knobs.h"
#include "ecgpu_msm_chunk.h"
#include "ecgpu_rows.h"

// -DECGPU_MSM_COMBINE_ROWS=0: the doublings of the k256 Horner chain on quad lanes as in round 4 (A/B: profiles/r05/msm_combine_rows_ab.txt)
#ifndef ECGPU_MSM_COMBINE_ROWS
#define ECGPU_MSM_COMBINE_ROWS 1
#endif

namespace ecgpu {

// ---- prepare ----------------------------------------------------------------------------------------------
// One lane per term.  The scalar is cut into MsmSplit<C, GLV>::SUB sub-scalars (ecgpu_recode.h: the folded scalar; in GLV
// mode, k256 only, the two halves, the second one against lambda P = (beta x, y)); sub-term h of term i has index j = h * npad + i
// (npad = n rounded up to a multiple of 64), so that a wave still writes 64 consecutive digits, points and validity bits
// per half.  Everything after this kernel sees nsub = SUB * npad independent entries.
// A digit is 16 bits: bucket | sign << 15 (all 2^16 codes are real at c = 16).  Whether sub-term j has a digit in
// window w at all (non-zero digit, finite point) is one bit of vmask[w][j / 64], written with a wave ballot.
// counts_a != nullptr (two-level sort): the level-A histogram — entries per (window, top `8` bits of the bucket) — is
// taken here as well, in dynamic LDS (nwin * npart counters, LDS atomics, one global atomicAdd per non-empty counter and
// workgroup): the sort then needs no pass of its own over the digits to count them.
template <class C, bool GLV>
__global__ void __launch_bounds__(BLOCK, C::N <= 12 ? 4 : 1)   // four waves per SIMD: k256 GLV took 129 registers around the reduction window (ecgpu_k256_reduce_asm.h)
k_msm_prepare(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points_xy,
              const uint8_t* __restrict__ points_inf, size_t n, size_t npad, int c, int nwin, uint32_t* __restrict__ pts,
              uint16_t* __restrict__ digits, unsigned long long* __restrict__ vmask, int* status,
              uint32_t* __restrict__ counts_a, int bits_b, int npart, int reps) {
    using G = Group<C>;
    using F = Field<C>;
    using S = MsmSplit<C, GLV>;
    constexpr int N = C::N;
    extern __shared__ uint32_t lds_count_a[];
    const int ncount = counts_a ? nwin * npart : 0;
    for (int t = threadIdx.x; t < ncount; t += BLOCK) lds_count_a[t] = 0;
    if (counts_a) __syncthreads();
#pragma unroll 1
    for (int rep = 0; rep < reps; rep++) {      // `reps` consecutive groups of BLOCK terms per workgroup: one flush of the counters for all
        const size_t i = ((size_t)blockIdx.x * reps + rep) * BLOCK + threadIdx.x;
        if (i - (threadIdx.x & 63) >= npad) break;                      // wave-uniform: nothing of this wave is inside the plan
        const bool active = i < n;
        const size_t ii = active ? i : 0;
        uint32_t k[N];
        bool finite = false;
        uint32_t sub[S::SUB][S::KW];
        bool flip[S::SUB];
        if (active) {
            load_scalar<C>(k, scalars, ii, status);
            S::split(k, sub, flip);
            Fe<C::NL> b = G::curve_b();
            Affine<C> a;
            uint32_t cx[N], cy[N];
            finite = load_affine_words<C>(&a, cx, cy, points_xy, points_inf, ii, b, status);
            if (finite) {
                if constexpr (F::REPR == REPR_U29_K256) {     // plain residues: the canonical words read ARE the packed form
                    store_words_vec<N>(pts + ii * (2 * N), cx);
                    store_words_vec<N>(pts + ii * (2 * N) + N, cy);
                } else {
                    store_packed_affine<C>(pts + ii * (2 * N), a.x, a.y);
                }
                if constexpr (S::SUB == 2) {                  // lambda P = (beta x, y): y's packed words are the ones just read
                    uint32_t bx[N];
                    F::pack(bx, F::mul(G::m(a.x), F::unpack(C::BETA)));
                    store_words_vec<N>(pts + (npad + ii) * (2 * N), bx);
                    if constexpr (F::REPR == REPR_U29_K256) {
                        store_words_vec<N>(pts + (npad + ii) * (2 * N) + N, cy);
                    } else {
                        F::pack(bx, G::m(a.y));
                        store_words_vec<N>(pts + (npad + ii) * (2 * N) + N, bx);
                    }
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < S::SUB; h++) {
                flip[h] = false;
#pragma unroll
                for (int t = 0; t < S::KW; t++) sub[h][t] = 0;
            }
        }
        const size_t nsub = S::SUB * npad, nmask = nsub / 64;
#pragma unroll
        for (int h = 0; h < S::SUB; h++) {
            const size_t j = (size_t)h * npad + i;
            MsmDigitStream<S::KW> ds;               // msm_digit's digits in window order, no dynamically indexed scalar word
            ds.init(sub[h]);
#pragma unroll 1
            for (int w = 0; w < nwin; w++) {
                MsmDigit d = ds.next(w, c, nwin, (uint32_t)j, flip[h], S::KBITS);
                const bool valid = active && finite && d.nonzero;
                if (active) digits[(size_t)w * nsub + j] = (uint16_t)(d.bucket | (d.neg << 15));
                unsigned long long m = __ballot(valid);
                if ((threadIdx.x & 63) == 0) vmask[(size_t)w * nmask + (j >> 6)] = m;
                if (counts_a && valid) atomicAdd(&lds_count_a[w * npart + (int)(d.bucket >> bits_b)], 1u);
            }
        }
    }
    if (counts_a) {
        __syncthreads();
        for (int t = threadIdx.x; t < ncount; t += BLOCK) {
            const uint32_t v = lds_count_a[t];
            if (v) atomicAdd(&counts_a[t], v);
        }
    }
}

// ---- counting sort ---------------------------------------------------------------------------------------------
// grid (ntiles, nwin), 1024 lanes, dynamic LDS = nb * 4 bytes: tile_hist[w][tile][b] = #terms of the tile in bucket b
static __global__ void __launch_bounds__(1024)
k_msm_hist(const uint16_t* __restrict__ digits, const unsigned long long* __restrict__ vmask, size_t n, size_t tile,
           size_t nb, uint32_t* __restrict__ tile_hist) {
    extern __shared__ uint32_t lds_hist[];
    const size_t w = blockIdx.y, t = blockIdx.x;
    for (size_t b = threadIdx.x; b < nb; b += blockDim.x) lds_hist[b] = 0;
    __syncthreads();
    size_t lo = t * tile, hi = lo + tile < n ? lo + tile : n;
    const uint16_t* dw = digits + w * n;
    const unsigned long long* vw = vmask + w * ((n + 63) / 64);
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {       // tiles start at multiples of 64
        if ((vw[i >> 6] >> (i & 63)) & 1) atomicAdd(&lds_hist[dw[i] & 0x7FFFu], 1u);
    }
    __syncthreads();
    uint32_t* out = tile_hist + (w * gridDim.x + t) * nb;
    for (size_t b = threadIdx.x; b < nb; b += blockDim.x) out[b] = lds_hist[b];
}

// one lane per (window, bucket): exclusive prefix over the tiles in place, bucket totals to counts
static __global__ void __launch_bounds__(BLOCK)
k_msm_tile_scan(uint32_t* __restrict__ tile_hist, size_t ntiles, size_t nb, int nwin, uint32_t* __restrict__ counts) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nb * nwin) return;
    size_t w = gid / nb, b = gid % nb;
    uint32_t run = 0;
    for (size_t t = 0; t < ntiles; t++) {
        uint32_t* p = tile_hist + (w * ntiles + t) * nb + b;
        uint32_t v = *p;
        *p = run;
        run += v;
    }
    counts[gid] = run;
}

// ---- scan: offsets[w][b] = sum_{b' < b} counts[w][b'] -------------------------------------------------------
// big_any != nullptr: big_any[window] = 1 if some count of the window exceeds big_limit, else 0 (the packed sort's level A:
// is there a partition for k_msm_sort_b_big?)
// copy: a second array that receives the same offsets (the scatter's cursors: no copy launch between the scan and the scatter)
static __global__ void __launch_bounds__(1024) k_msm_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                          size_t nb, uint32_t big_limit = 0, uint32_t* __restrict__ big_any = nullptr,
                                                          uint32_t* __restrict__ copy = nullptr) {
    __shared__ uint32_t part[1024];
    const uint32_t* cw = counts + (size_t)blockIdx.x * nb;
    uint32_t* ow = offsets + (size_t)blockIdx.x * nb;
    size_t per = (nb + 1023) / 1024;
    size_t lo = (size_t)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    if (lo > nb) lo = nb;
    uint32_t s = 0;
    int over = 0;
    for (size_t j = lo; j < hi; j++) {
        s += cw[j];
        over |= cw[j] > big_limit;
    }
    if (big_any != nullptr) {
        over = __syncthreads_or(over);
        if (threadIdx.x == 0) big_any[blockIdx.x] = over ? 1u : 0u;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan of the partials
        uint32_t v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    uint32_t* cp = copy != nullptr ? copy + (size_t)blockIdx.x * nb : nullptr;
    for (size_t j = lo; j < hi; j++) {
        ow[j] = run;
        if (cp != nullptr) cp[j] = run;
        run += cw[j];
    }
}

// grid (ntiles, nwin), 1024 lanes, dynamic LDS = nb * 4: scatter the tile's terms to their bucket runs
static __global__ void __launch_bounds__(1024)
k_msm_scatter(const uint16_t* __restrict__ digits, const unsigned long long* __restrict__ vmask, size_t n, size_t tile,
              size_t nb, const uint32_t* __restrict__ tile_hist, const uint32_t* __restrict__ offsets,
              uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t lds_cursor[];
    const size_t w = blockIdx.y, t = blockIdx.x;
    const uint32_t* th = tile_hist + (w * gridDim.x + t) * nb;
    const uint32_t* ow = offsets + w * nb;
    for (size_t b = threadIdx.x; b < nb; b += blockDim.x) lds_cursor[b] = ow[b] + th[b];
    __syncthreads();
    size_t lo = t * tile, hi = lo + tile < n ? lo + tile : n;
    const uint16_t* dw = digits + w * n;
    uint32_t* sw = sorted + w * n;
    const unsigned long long* vw = vmask + w * ((n + 63) / 64);
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        if ((vw[i >> 6] >> (i & 63)) & 1) {
            uint16_t d = dw[i];
            uint32_t pos = atomicAdd(&lds_cursor[d & 0x7FFFu], 1u);
            sw[pos] = (uint32_t)i | ((uint32_t)(d >> 15) << 31);
        }
    }
}

// ---- two-level sort for large MSMs ------------------------------------------------------------------------------------
// The single-level scatter above issues one 4-byte store per entry to an address that is random within the window's
// 64 MiB run: 2.7x10^8 separate L2 write requests at 2^24 terms, 4 of the kernel's 4.7 ms (measured: the same kernel
// without its stores takes 0.76 ms).  Here a workgroup takes a tile of 8192 entries, ranks them by key with LDS
// atomics, lays them out key-sorted in LDS and writes the runs of equal keys as consecutive addresses, so that a wave's
// store covers a few cache lines instead of 64.  That needs runs much longer than the 8 entries per (tile, bucket) a
// 2^15-bucket sort produces, hence two levels: A partitions by the top 8 bits of the bucket (256 keys: 32 entries per
// run), B sorts by the remaining bits inside each partition — a tile of the partitioned array touches one or two
// partitions, which it handles one after the other (one round per partition present, <= 128 keys each).
// Where a run starts comes from a global cursor per key, advanced by one atomicAdd per (tile, key): the order of the
// entries inside a bucket depends on the scheduling of the workgroups, which bucket sums do not care about.
constexpr int MSM_SORT2_TILE = 8192;          // entries per workgroup
constexpr int MSM_SORT2_PER_LANE = MSM_SORT2_TILE / 1024;
constexpr int MSM_SORT2_BITS_A = 8;

struct MsmSort2Src {
    const uint16_t* codes;                  // [nwin][n] digit codes (bucket | sign << 15): term order (level A) / partitioned (B)
    const unsigned long long* vmask;        // level A: validity bits [nwin][ceil(n/64)]
    const uint32_t* idx;                    // level B: term index of entry j
    const uint32_t* part_offsets;           // level B: [nwin][npart]; the window holds last offset + last count entries
    const uint32_t* part_counts;
    size_t npart;
};

// LEVEL_B = false: key = bucket >> bits_b (one round).  LEVEL_B = true: one round per partition (bucket >> bits_b) present
// in the tile, key = bucket & (2^bits_b - 1).  Counter / cursor index: level A the key, level B the bucket.
// COUNT_ONLY: gcnt[w][index] += number of entries (histogram pass).  Otherwise gcnt holds the cursors (initialised to the
// run starts) and the entries are written: level A out_idx = term, out_key = code; level B out_idx = term | sign << 31.
template <bool LEVEL_B, bool COUNT_ONLY>
static __global__ void __launch_bounds__(1024, 8)       // two workgroups per CU: 64 VGPRs (level B needed 80 and ran one per CU)
k_msm_sort2(MsmSort2Src src, size_t n, int bits_b, size_t nindex, uint32_t* __restrict__ gcnt, uint32_t* __restrict__ out_idx,
            uint16_t* __restrict__ out_key) {
    __shared__ uint32_t cnt[256], loc[256], gbase[256], wtot[4];
    __shared__ uint32_t stage[COUNT_ONLY ? 1 : MSM_SORT2_TILE];
    __shared__ uint16_t stage_k[COUNT_ONLY ? 1 : MSM_SORT2_TILE];
    const size_t w = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    size_t tot = n;
    if (LEVEL_B) tot = src.part_offsets[w * src.npart + src.npart - 1] + src.part_counts[w * src.npart + src.npart - 1];
    const size_t lo = (size_t)blockIdx.x * MSM_SORT2_TILE;
    if (lo >= tot) return;
    const size_t hi = lo + MSM_SORT2_TILE < tot ? lo + MSM_SORT2_TILE : tot;
    const uint16_t* cw = src.codes + w * n;
    const uint32_t nkeys = LEVEL_B ? 1u << bits_b : (uint32_t)nindex;
    const uint32_t mask_b = (1u << bits_b) - 1;
    // per entry: the 16-bit code in the low half, its rank among the tile's entries of the same key in the high half (filled in
    // below; < 8192); 0xFFFFFFFF: no entry.  One register per entry instead of two: level B fits 64 VGPRs, i.e. two workgroups
    // per CU, without spilling.
    uint32_t code[MSM_SORT2_PER_LANE], term[MSM_SORT2_PER_LANE];
#pragma unroll
    for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
        const size_t i = lo + (size_t)u * 1024 + tid;                 // tiles start at multiples of 64
        bool ok = i < hi;
        if (!LEVEL_B && ok) ok = (src.vmask[w * ((n + 63) / 64) + (i >> 6)] >> (i & 63)) & 1;
        code[u] = ok ? (uint32_t)cw[i] : 0xFFFFFFFFu;
        term[u] = LEVEL_B ? (ok ? src.idx[w * n + i] : 0u) : (uint32_t)i;
    }
    uint32_t r_lo = 0, r_hi = 0;
    if (LEVEL_B) {                                                      // the tile is sorted by partition
        r_lo = (uint32_t)(cw[lo] & 0x7FFFu) >> bits_b;
        r_hi = (uint32_t)(cw[hi - 1] & 0x7FFFu) >> bits_b;
    }
    for (uint32_t r = r_lo; r <= r_hi; r++) {
        if (tid < 256) cnt[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            if (code[u] == 0xFFFFFFFFu) continue;
            const uint32_t b = code[u] & 0x7FFFu;
            if (LEVEL_B && (b >> bits_b) != r) continue;
            const uint32_t rank = atomicAdd(&cnt[LEVEL_B ? b & mask_b : b >> bits_b], 1u);
            code[u] = (code[u] & 0xFFFFu) | (rank << 16);
        }
        __syncthreads();
        const uint32_t mine = tid < nkeys ? cnt[tid] : 0;
        const size_t gi = w * nindex + (LEVEL_B ? ((size_t)r << bits_b) + tid : (size_t)tid);
        if (COUNT_ONLY) {
            if (mine) atomicAdd(&gcnt[gi], mine);
            __syncthreads();
            continue;
        }
        // The run starts come from a device-scope atomic on a cursor every tile of the window hammers: its round trip
        // (microseconds) is the longest single wait of the tile.  It is issued here and its result is only stored to LDS
        // after the scan and the staging of the entries, so that it flies under them instead of in front of them.
        uint32_t gb = 0;
        if (mine) gb = atomicAdd(&gcnt[gi], mine);
        // exclusive scan of the (<= 256) counts: inside each of the first four waves by lane shuffles, then the four wave
        // totals (two barriers instead of the nineteen of a scan through LDS)
        uint32_t incl = mine;
        if (tid < 256) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if ((int)(tid & 63) >= d) incl += v;
            }
            if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        }
        __syncthreads();
        const uint32_t total_r = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        if (tid < 256) {
            uint32_t before = 0;
            for (uint32_t k = 0; k < (tid >> 6); k++) before += wtot[k];
            loc[tid] = before + incl - mine;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            if (code[u] == 0xFFFFFFFFu) continue;
            const uint32_t b = code[u] & 0x7FFFu;
            if (LEVEL_B && (b >> bits_b) != r) continue;
            const uint32_t slot = loc[LEVEL_B ? b & mask_b : b >> bits_b] + (code[u] >> 16);
            stage[slot] = LEVEL_B ? term[u] | (((code[u] >> 15) & 1u) << 31) : term[u];
            stage_k[slot] = (uint16_t)code[u];
        }
        if (mine) gbase[tid] = gb;
        __syncthreads();
        for (uint32_t slot = tid; slot < total_r; slot += 1024) {
            const uint32_t c16 = stage_k[slot];
            const uint32_t k = LEVEL_B ? c16 & mask_b : (c16 & 0x7FFFu) >> bits_b;
            const size_t dst = w * n + gbase[k] + (slot - loc[k]);
            out_idx[dst] = stage[slot];
            if (!LEVEL_B) out_key[dst] = (uint16_t)c16;
        }
        __syncthreads();
    }
}

// ---- two-level sort, packed form (round 4) ---------------------------------------------------------------------------------
// The same two levels with ONE 32-bit word per entry between them and no global atomics in level B:
//   level A  (k_msm_sort_a) partitions a window's entries by the top bits_a bits of the bucket and writes
//            packed = index | sign << idx_bits | (bucket & (2^bits_b - 1)) << (idx_bits + 1)
//            — 4 bytes per entry instead of a 4-byte index + a 2-byte code (idx_bits = ceil(log2(entries per window)); the
//            plan uses this form whenever idx_bits + 1 + bits_b <= 32).  A lane takes 8 CONSECUTIVE entries: their codes are
//            one 16-byte load and their validity bits one byte, instead of 8 halfword loads + 8 looks at the mask words;
//   level B  (k_msm_sort_b) is ONE workgroup per (partition, window): it counts the partition's entries per bucket in an LDS
//            histogram (pass 1), turns the counts into the window's `counts` / `offsets` rows itself (the partition's start
//            comes from level A's scan), and scatters tile by tile through the LDS staging buffer with LDS cursors (pass 2).
//            No separate counting launch, no scan over 2^15 counters, no cursor copy, no global atomic per (tile, key).
// A partition is ~65,536 entries (256 KiB) at 2^24 terms.  A degenerate input (all scalars equal, all ones, bit vectors) puts a
// whole window into ONE partition; one workgroup streaming 2^24 entries would take 15 ms.  Partitions above
// msm_sortb_big_limit() entries (four times the mean) are therefore left out by k_msm_sort_b and sorted the round-3 way by
// k_msm_sort_b_big: tiles of 8192 positions spread over the whole chip, a counting pass with one global atomicAdd per
// (tile, bucket), a scan per big partition, a scatter pass with global cursors; the lanes of those tiles all want the same
// few LDS counters and rank with one LDS atomic per distinct key and wave (msm_lds_rank<true>).  With random scalars the three
// extra launches find nothing to do and exit (~0.02 ms at 2^24 terms).
constexpr int MSM_SORTP_MAX_BITS_A = 9;        // level-A keys: <= 512 (k_msm_prepare holds nwin x 2^bits_a counters in LDS)
constexpr int MSM_SORTP_MAX_BITS_B = 8;        // level-B keys: <= 256
// partitions larger than this go to k_msm_sort_b_big (ne entries per window in npart partitions)
inline uint32_t msm_sortb_big_limit(size_t ne, size_t npart) {
    const size_t lim = 4 * (ne / npart);
    return (uint32_t)(lim < 32768 ? 32768 : lim);
}

// rank of this lane's entry among the entries of its key counted so far in cnt[] (LDS); AGG: the lanes of a wave that
// share a key are counted with one atomic per distinct key (loop over the distinct keys of the wave)
template <bool AGG>
__device__ __forceinline__ uint32_t msm_lds_rank(uint32_t* cnt, uint32_t key, bool live) {
    if constexpr (!AGG) {
        return live ? atomicAdd(&cnt[key], 1u) : 0u;
    } else {
        uint32_t rank = 0;
        const int lane = (int)(threadIdx.x & 63);
        unsigned long long todo = __ballot(live);
        while (todo) {                                   // wave-uniform
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lk = (uint32_t)__shfl((int)key, leader, 64);
            const bool same = live && key == lk;
            const unsigned long long m = __ballot(same);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&cnt[lk], (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (same) rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        return rank;
    }
}

// loc[0 .. nkeys) = exclusive prefix sums of cnt[0 .. nkeys), nkeys <= blockDim.x (a multiple of 64, <= 1024); returns the
// total.  Every thread calls it; cnt must be final (barrier before), loc is valid on return (barrier inside).
__device__ __forceinline__ uint32_t msm_block_scan(const uint32_t* cnt, uint32_t* loc, uint32_t nkeys, uint32_t* wtot) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t mine = tid < nkeys ? cnt[tid] : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((int)(tid & 63) >= d) incl += v;
    }
    if ((tid & 63) == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t k = 0; k < nwaves; k++) {
        const uint32_t v = wtot[k];
        if (k < wave) before += v;
        total += v;
    }
    if (tid < nkeys) loc[tid] = before + incl - mine;
    __syncthreads();
    return total;
}

// grid (ceil(n / 8192), nwin), 1024 lanes.  cursor[w][key]: the run starts (level A's scan), advanced by one global
// atomicAdd per (tile, key) as in k_msm_sort2.
static __global__ void __launch_bounds__(1024, 8)
k_msm_sort_a(const uint16_t* __restrict__ digits, const unsigned long long* __restrict__ vmask, size_t n, int bits_b, int idx_bits,
             uint32_t npart, uint32_t* __restrict__ cursor, uint32_t* __restrict__ out) {
    __shared__ uint32_t cnt[1 << MSM_SORTP_MAX_BITS_A], loc[1 << MSM_SORTP_MAX_BITS_A], gbase[1 << MSM_SORTP_MAX_BITS_A], wtot[16];
    __shared__ uint32_t stage[MSM_SORT2_TILE];
    __shared__ uint16_t stage_k[MSM_SORT2_TILE];
    const size_t w = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * MSM_SORT2_TILE + (size_t)tid * MSM_SORT2_PER_LANE;   // n is a multiple of 64
    const bool in = i0 < n;
    uint32_t cw[4] = {0, 0, 0, 0};
    uint32_t vb = 0;
    if (in) {
        const uint4 dv = *reinterpret_cast<const uint4*>(digits + w * n + i0);
        cw[0] = dv.x; cw[1] = dv.y; cw[2] = dv.z; cw[3] = dv.w;
        vb = reinterpret_cast<const uint8_t*>(vmask + w * (n / 64))[i0 >> 3];
    }
    if (tid < npart) cnt[tid] = 0;
    __syncthreads();
    const uint32_t mask_b = (1u << bits_b) - 1;
    uint32_t key[MSM_SORT2_PER_LANE], rank[MSM_SORT2_PER_LANE];
#pragma unroll
    for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
        const uint32_t c16 = (cw[u >> 1] >> (16 * (u & 1))) & 0xFFFFu;
        key[u] = (c16 & 0x7FFFu) >> bits_b;
        rank[u] = msm_lds_rank<false>(cnt, key[u], (vb >> u) & 1u);
    }
    __syncthreads();
    const uint32_t mine = tid < npart ? cnt[tid] : 0u;
    uint32_t gb = 0;
    if (mine) gb = atomicAdd(&cursor[w * npart + tid], mine);          // flies under the scan and the staging
    const uint32_t total = msm_block_scan(cnt, loc, npart, wtot);
#pragma unroll
    for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
        if (!((vb >> u) & 1u)) continue;
        const uint32_t c16 = (cw[u >> 1] >> (16 * (u & 1))) & 0xFFFFu;
        const uint32_t slot = loc[key[u]] + rank[u];
        stage[slot] = (uint32_t)(i0 + u) | ((c16 >> 15) << idx_bits) | ((c16 & mask_b) << (idx_bits + 1));
        stage_k[slot] = (uint16_t)key[u];
    }
    if (mine) gbase[tid] = gb;
    __syncthreads();
    uint32_t* ow = out + w * n;
    for (uint32_t slot = tid; slot < total; slot += 1024) {
        const uint32_t k = stage_k[slot];
        ow[gbase[k] + (slot - loc[k])] = stage[slot];
    }
}

// grid (npart, nwin); blockDim 256 or 1024.  in: level A's output; part_offsets / part_counts [nwin][npart]: level A's
// scan / histogram.  Writes counts / offsets [nwin][npart << bits_b] and sorted[w][.] = index | sign << 31.
__device__ __forceinline__ void msm_sort_b_body(const uint32_t* __restrict__ src, uint32_t np, uint32_t lo, int bits_b, int idx_bits,
                                                uint32_t* __restrict__ counts_row, uint32_t* __restrict__ offsets_row,
                                                uint32_t* __restrict__ sorted_w, uint32_t* hist, uint32_t* cur, uint32_t* cnt,
                                                uint32_t* loc, uint32_t* wtot, uint32_t* stage) {
    const uint32_t tid = threadIdx.x, T = blockDim.x, tile = T * MSM_SORT2_PER_LANE;
    const uint32_t nkeys = 1u << bits_b, kshift = (uint32_t)idx_bits + 1, idx_mask = (1u << idx_bits) - 1;
    if (tid < nkeys) hist[tid] = 0;
    __syncthreads();
    constexpr int P1 = 2 * MSM_SORT2_PER_LANE;                         // pass 1 keeps no ranks: twice the loads in flight per lane
    for (uint32_t base = 0; base < np; base += T * P1) {               // pass 1: the partition's histogram
        uint32_t v[P1];
#pragma unroll
        for (int u = 0; u < P1; u++) {
            const uint32_t i = base + (uint32_t)u * T + tid;
            v[u] = src[i < np ? i : np - 1];                           // (no exec-masked load: the dead lanes re-read the last entry)
        }
#pragma unroll
        for (int u = 0; u < P1; u++) {
            const uint32_t i = base + (uint32_t)u * T + tid;
            (void)msm_lds_rank<false>(hist, v[u] >> kshift, i < np);
        }
    }
    __syncthreads();
    (void)msm_block_scan(hist, loc, nkeys, wtot);
    if (tid < nkeys) {
        counts_row[tid] = hist[tid];
        offsets_row[tid] = lo + loc[tid];
        cur[tid] = lo + loc[tid];
    }
    for (uint32_t base = 0; base < np; base += tile) {                 // pass 2: tile by tile through the staging buffer
        if (tid < nkeys) cnt[tid] = 0;
        __syncthreads();                                               // (also: cur[] of the previous tile is final)
        uint32_t v[MSM_SORT2_PER_LANE], rank[MSM_SORT2_PER_LANE];
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = base + (uint32_t)u * T + tid;
            v[u] = src[i < np ? i : np - 1];                           // (no exec-masked load: the dead lanes re-read the last entry)
        }
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = base + (uint32_t)u * T + tid;
            rank[u] = msm_lds_rank<false>(cnt, v[u] >> kshift, i < np);
        }
        __syncthreads();
        const uint32_t total = msm_block_scan(cnt, loc, nkeys, wtot);
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = base + (uint32_t)u * T + tid;
            if (i < np) stage[loc[v[u] >> kshift] + rank[u]] = v[u];
        }
        __syncthreads();
        for (uint32_t slot = tid; slot < total; slot += T) {
            const uint32_t p = stage[slot], k = p >> kshift;
            sorted_w[cur[k] + (slot - loc[k])] = (p & idx_mask) | (((p >> idx_bits) & 1u) << 31);
        }
        __syncthreads();
        if (tid < nkeys) cur[tid] += cnt[tid];
    }
}
static __global__ void __launch_bounds__(1024, 8)
k_msm_sort_b(const uint32_t* __restrict__ in, size_t n, int bits_b, int idx_bits, uint32_t npart, uint32_t big_limit,
             const uint32_t* __restrict__ part_offsets, const uint32_t* __restrict__ part_counts, uint32_t* __restrict__ counts,
             uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t hist[1 << MSM_SORTP_MAX_BITS_B], cur[1 << MSM_SORTP_MAX_BITS_B], cnt[1 << MSM_SORTP_MAX_BITS_B],
        loc[1 << MSM_SORTP_MAX_BITS_B], wtot[16];
    __shared__ uint32_t stage[MSM_SORT2_TILE];
    const size_t w = blockIdx.y, p = blockIdx.x;
    const uint32_t lo = part_offsets[w * npart + p], np = part_counts[w * npart + p];
    const size_t nb = (size_t)npart << bits_b, row = w * nb + (p << bits_b);
    if (np > big_limit) {                              // left to k_msm_sort_b_big, whose counting pass adds into zeroed counters
        if (threadIdx.x < (1u << bits_b)) counts[row + threadIdx.x] = 0;
        return;
    }
    msm_sort_b_body(in + w * n + lo, np, lo, bits_b, idx_bits, counts + row, offsets + row, sorted + w * n, hist, cur, cnt, loc, wtot,
                    stage);
}

// The big partitions (see above).  grid (tiles of 8192 POSITIONS of the window's level-A array, nwin), 1024 lanes; a tile handles
// the big partitions it overlaps one after the other and ignores the others.  COUNT_ONLY: counts[w][bucket] += the tile's entries
// (zeroed by k_msm_sort_b).  Otherwise `cursor` holds the run starts (k_msm_sort_b_big_offsets) and the entries are written.
template <bool COUNT_ONLY>
static __global__ void __launch_bounds__(1024, 8)
k_msm_sort_b_big(const uint32_t* __restrict__ in, size_t n, int bits_b, int idx_bits, uint32_t npart, uint32_t big_limit,
                 const uint32_t* __restrict__ part_offsets, const uint32_t* __restrict__ part_counts,
                 const uint32_t* __restrict__ big_any, uint32_t* __restrict__ gcnt, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[1 << MSM_SORTP_MAX_BITS_B], loc[1 << MSM_SORTP_MAX_BITS_B], gbase[1 << MSM_SORTP_MAX_BITS_B], wtot[16];
    __shared__ uint32_t stage[COUNT_ONLY ? 1 : MSM_SORT2_TILE];
    const size_t w = blockIdx.y;
    if (!big_any[w]) return;                            // the window has no big partition (k_msm_scan): the normal case
    const uint32_t tid = threadIdx.x;
    const uint32_t* off = part_offsets + w * npart;
    const uint32_t* pc = part_counts + w * npart;
    const uint32_t tot = off[npart - 1] + pc[npart - 1];
    for (uint32_t lo = blockIdx.x * (uint32_t)MSM_SORT2_TILE; lo < tot; lo += gridDim.x * (uint32_t)MSM_SORT2_TILE) {
    const uint32_t hi = lo + MSM_SORT2_TILE < tot ? lo + MSM_SORT2_TILE : tot;
    // first partition that reaches past `lo` (partitions are consecutive; empty ones have zero length)
    uint32_t a = 0, b = npart;                          // invariant: every partition < a ends at or before lo
    while (a < b) {
        const uint32_t m = (a + b) / 2;
        if (off[m] + pc[m] <= lo) a = m + 1;
        else b = m;
    }
    const uint32_t nkeys = 1u << bits_b, kshift = (uint32_t)idx_bits + 1, idx_mask = (1u << idx_bits) - 1;
    const size_t nb = (size_t)npart << bits_b;
    for (uint32_t r = a; r < npart && off[r] < hi; r++) {
        const uint32_t np = pc[r];
        if (np <= big_limit) continue;                                  // (wave-uniform)
        const uint32_t s0 = off[r] > lo ? off[r] : lo, s1 = off[r] + np < hi ? off[r] + np : hi;
        if (tid < nkeys) cnt[tid] = 0;
        __syncthreads();
        uint32_t v[MSM_SORT2_PER_LANE], rank[MSM_SORT2_PER_LANE];
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = s0 + (uint32_t)u * 1024 + tid;
            v[u] = in[w * n + (i < s1 ? i : s1 - 1)];
        }
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = s0 + (uint32_t)u * 1024 + tid;
            rank[u] = msm_lds_rank<true>(cnt, v[u] >> kshift, i < s1);
        }
        __syncthreads();
        const uint32_t mine = tid < nkeys ? cnt[tid] : 0u;
        const size_t gi = w * nb + ((size_t)r << bits_b) + tid;
        if (COUNT_ONLY) {
            if (mine) atomicAdd(&gcnt[gi], mine);
            __syncthreads();
            continue;
        }
        uint32_t gb = 0;
        if (mine) gb = atomicAdd(&gcnt[gi], mine);
        const uint32_t total = msm_block_scan(cnt, loc, nkeys, wtot);
#pragma unroll
        for (int u = 0; u < MSM_SORT2_PER_LANE; u++) {
            const uint32_t i = s0 + (uint32_t)u * 1024 + tid;
            if (i < s1) stage[loc[v[u] >> kshift] + rank[u]] = v[u];
        }
        if (mine) gbase[tid] = gb;
        __syncthreads();
        for (uint32_t slot = tid; slot < total; slot += 1024) {
            const uint32_t p = stage[slot], k = p >> kshift;
            sorted[w * n + gbase[k] + (slot - loc[k])] = (p & idx_mask) | (((p >> idx_bits) & 1u) << 31);
        }
        __syncthreads();
    }
    }
}

// grid (npart, nwin), 256 lanes: offsets and cursors of the big partitions from their counted buckets
static __global__ void __launch_bounds__(256)
k_msm_sort_b_big_offsets(int bits_b, uint32_t npart, uint32_t big_limit, const uint32_t* __restrict__ part_offsets,
                         const uint32_t* __restrict__ part_counts, const uint32_t* __restrict__ big_any,
                         const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t c[1 << MSM_SORTP_MAX_BITS_B], loc[1 << MSM_SORTP_MAX_BITS_B], wtot[16];
    const size_t w = blockIdx.y;
    if (!big_any[w]) return;
    for (size_t p = blockIdx.x; p < npart; p += gridDim.x) {
    if (part_counts[w * npart + p] <= big_limit) continue;
    const uint32_t lo = part_offsets[w * npart + p], nkeys = 1u << bits_b, tid = threadIdx.x;
    const size_t row = (w * npart + p) << bits_b;
    if (tid < nkeys) c[tid] = counts[row + tid];
    __syncthreads();
    (void)msm_block_scan(c, loc, nkeys, wtot);
    if (tid < nkeys) {
        offsets[row + tid] = lo + loc[tid];
        cursor[row + tid] = lo + loc[tid];
    }
    __syncthreads();
    }
}

// ---- accumulate: the hot loop --------------------------------------------------------------------------------------
template <class C>
struct MsmPointsHbm {
    const uint32_t* pts;
    __device__ void load(PackedPoint<2 * C::N>& p, uint32_t term) const {
        load_words_vec<2 * C::N>(p.w, pts + (size_t)term * (2 * C::N));
    }
};
template <class C>
struct MsmPartialsHbm {       // [slot][4] raw elements: X, Y, ZZ, ZZZ
    uint32_t* base;
    __device__ void put(size_t slot, const Xyzz<C>& p) {
        constexpr int NS = Field<C>::NS;
        uint32_t* d = base + slot * (4 * NS);
        store_raw<C>(d, p.x);
        store_raw<C>(d + NS, p.y);
        store_raw<C>(d + 2 * NS, p.zz);
        store_raw<C>(d + 3 * NS, p.zzz);
    }
    __device__ Xyzz<C> get(size_t slot) const {
        constexpr int NS = Field<C>::NS;
        const uint32_t* s = base + slot * (4 * NS);
        Xyzz<C> p;
        p.x = load_raw<C>(s);
        p.y = load_raw<C>(s + NS);
        p.zz = load_raw<C>(s + 2 * NS);
        p.zzz = load_raw<C>(s + 3 * NS);
        return p;
    }
};

// one lane per (window, chunk)
// (-DECGPU_MSM_ACC_WAVES=4, an A/B knob of the build: 128 registers + 108 bytes of scratch per lane for k256 — measured 10 % slower,
// 15.8 against 14.4 ms at 2^24 terms, profiles/r04/msm_accumulate_four_waves_ab.txt)
#ifndef ECGPU_MSM_ACC_WAVES
#define ECGPU_MSM_ACC_WAVES 3
#endif
static_assert(ECGPU_MSM_ACC_WAVES >= 1 && ECGPU_MSM_ACC_WAVES <= 4,
              "the k256 reduction's assembly blocks own v[94:127] (ecgpu_k256_reduce_asm.h): a kernel that includes them needs 128 VGPRs, "
              "i.e. at most four waves per SIMD");
template <class C>
__global__ void __launch_bounds__(64, C::N <= 8 ? ECGPU_MSM_ACC_WAVES : C::N <= 12 ? 2 : 1)
k_msm_accumulate(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ sorted,
                 const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, size_t n, size_t nb,
                 int nwin, size_t chunk, size_t nchunks, uint32_t* __restrict__ partials, uint32_t* __restrict__ zero_word) {
    using G = Group<C>;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0 && zero_word != nullptr) *zero_word = 0;       // the counter of k_msm_bucket_finish's list (no memset launch in the chain)
    if (gid >= nchunks * nwin) return;
    size_t w = gid / nchunks, q = gid % nchunks;
    const uint32_t* ow = offsets + w * nb;
    const uint32_t total = ow[nb - 1] + counts[w * nb + nb - 1];
    MsmPointsHbm<C> points{pts};
    MsmPartialsHbm<C> sink{partials + w * (nb + nchunks) * (4 * Field<C>::NS)};
    msm_chunk_accumulate<C>(sorted + w * n, ow, total, (uint32_t)nb, (uint32_t)chunk, (uint32_t)q, G::curve_b(), points, sink);
}

// -DECGPU_MSM_FUSED_TAIL=0: bucket finish and running sums as the two launches of round 4 (A/B: profiles/r05/)
#ifndef ECGPU_MSM_FUSED_TAIL
#define ECGPU_MSM_FUSED_TAIL 1
#endif
// The fused tail kernel (k_msm_finish_segments: the bucket finish inside the running sums — no launch and no round trip of the
// bucket sums) is built and correct but NOT the default: measured on MI355X (profiles/r05/msm_tail_fused_ab.txt), k256, before
// the reduction went to assembly: 0.365 against 0.391 ms at 2^24 terms, 0.32 against 0.27 at 2^21 GLV terms (65,536 segment lanes
// = one wave per SIMD, every lane a chain of 4 x ~3 stretches + 29 point operations; a lane per BUCKET is four times the
// parallelism for the stretch sums).  With the assembly reduction k_msm_bucket_finish needs 168 registers instead of 256 + scratch
// and runs three waves per SIMD: the two launches then win at both sizes (0.49 against 0.64 ms at 2^24, 0.37 against 0.40 at 2^21
// on a box with slow tail kernels).  ECGPU_MSM_FUSED_TAIL=1 selects the fused form (sets up to 384 bits: the twenty-limb p521 does
// not fit its registers).
template <class C>
inline bool msm_fused_tail(const MsmPlan& p) {
    (void)p;
    if (const char* e = knob("ECGPU_MSM_FUSED_TAIL")) return e[0] == '1' && ECGPU_MSM_FUSED_TAIL != 0 && C::N <= 12;
    return false;
}

// A bucket normally has one or two partial sums.  A degenerate input (all scalars equal, all ones) gives ONE bucket
// per window thousands of them; such buckets are handed to a whole workgroup each (k_msm_big_buckets) instead of
// being walked by a single lane.
constexpr uint32_t MSM_BIG_PARTIALS = 32;

// one lane per (window, bucket); big_list[0] = number of deferred buckets, big_list[1..] their ids.
// Two waves per SIMD for the 256-bit sets (k256: 256 registers, three of them in scratch — left to itself the compiler takes 258
// and ONE wave per SIMD: 0.137 -> 0.157 / 0.179 ms at 2^21 / 2^24 terms, profiles/r04/msm_tree_quad_lanes.txt), the whole register file
// for the wider ones.  Never three: at three waves
// per SIMD 83 registers went to scratch and every lane paid ~100 scratch round trips (0.13 ms for a kernel whose arithmetic
// is 10 us; profiles/r03/).
// (A second build of these tail kernels scheduled for instruction-level parallelism, `-amdgpu-sched-strategy=max-ilp`, was
// measured and dropped: no difference beyond noise, profiles/r03/msm_tail_variants_88130fd.txt.)
template <class C>
__global__ void __launch_bounds__(64, C::N <= 8 ? 2 : 1)
k_msm_bucket_finish(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                    const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ pts,
                    const uint32_t* __restrict__ sorted, size_t n, size_t nb, int nwin, size_t chunk, size_t nchunks,
                    uint32_t* __restrict__ buckets, uint32_t* __restrict__ big_list, uint32_t max_big) {
    using G = Group<C>;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nb * nwin) return;
    size_t w = gid / nb, b = gid % nb;
    const uint32_t first = offsets[gid], cnt = counts[gid];
    if (cnt != 0 && (first + cnt - 1) / (uint32_t)chunk - first / (uint32_t)chunk >= MSM_BIG_PARTIALS) {
        uint32_t slot = atomicAdd(big_list, 1u);
        if (slot < max_big) {                                 // (always: max_big is an upper bound)
            big_list[1 + slot] = (uint32_t)gid;
            return;
        }
    }
    MsmPartialsHbm<C> src{const_cast<uint32_t*>(partials) + w * (nb + nchunks) * (4 * Field<C>::NS)};
    MsmPointsHbm<C> points{pts};
    store_proj<C>(buckets, gid, msm_bucket_finish<C>((uint32_t)b, first, cnt, (uint32_t)chunk, G::curve_b(), src, sorted + w * n, points));
}

// one workgroup per deferred bucket: strided sums of its partials + an LDS tree
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_msm_big_buckets(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                  const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ pts,
                  const uint32_t* __restrict__ sorted, size_t n, size_t nb, size_t chunk, size_t nchunks,
                  uint32_t* __restrict__ buckets, const uint32_t* __restrict__ big_list, uint32_t max_big) {
    using G = Group<C>;
    __shared__ uint32_t lds[BLOCK * 3 * C::NL];
    // the launch is a fixed MSM_BIG_GRID workgroups that deal the listed buckets out among themselves: for random scalars the
    // list is empty and 4 x MSM_BIG_GRID waves leave on their first load (round 4 launched one workgroup per POSSIBLE list
    // entry: 75,968 empty waves at 2^24 terms)
    // (the writers count every candidate but store only the first max_big: the list is never read past what was written)
    const uint32_t nbig = big_list[0] < max_big ? big_list[0] : max_big;
    for (uint32_t it = blockIdx.x; it < nbig; it += gridDim.x) {
        const size_t gid = big_list[1 + it];
        const size_t w = gid / nb, b = gid % nb;
        const uint32_t first = offsets[gid], cnt = counts[gid];
        const uint32_t q0 = first / (uint32_t)chunk, q1 = (first + cnt - 1) / (uint32_t)chunk;
        MsmPartialsHbm<C> src{const_cast<uint32_t*>(partials) + w * (nb + nchunks) * (4 * Field<C>::NS)};
        MsmPointsHbm<C> points{pts};
        const Fe<C::NL> cb = G::curve_b();
        Proj<C> acc = G::identity();
        for (uint32_t q = q0 + threadIdx.x; q <= q1; q += BLOCK)
            acc = G::add(acc, msm_stretch_of<C>((uint32_t)b, q, first, cnt, (uint32_t)chunk, cb, src, sorted + w * n, points), cb);
        acc = block_sum<C>(acc, lds, cb);
        if (threadIdx.x == 0) store_proj<C>(buckets, gid, acc);
        __syncthreads();
    }
}
constexpr unsigned MSM_BIG_GRID = 256;

// the buckets k_msm_big_buckets takes, listed BEFORE the accumulation (counts and offsets are all it needs): one lane per
// (window, bucket).  For the fused tail below, which has no per-bucket launch of its own to do the listing in.
static __global__ void __launch_bounds__(256)
k_msm_find_big(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, size_t nbk, uint32_t chunk,
               uint32_t* __restrict__ big_list, uint32_t max_big) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nbk) return;
    const uint32_t first = offsets[gid], cnt = counts[gid];
    if (cnt != 0 && (first + cnt - 1) / chunk - first / chunk >= MSM_BIG_PARTIALS) {
        const uint32_t slot = atomicAdd(big_list, 1u);
        if (slot < max_big) big_list[1 + slot] = (uint32_t)gid;     // (always: max_big is an upper bound)
    }
}

// ---- a = 0 (k256): one COMPLETE projective doubling spread over the four lanes of a quad -----------------------------------
// Renes–Costello–Batina's doubling for a = 0 (the formulas of Group::dbl_a0: X3 = 2 XY (Y^2 - 9b Z^2), Y3 = 24b Y^2 Z^2 +
// (Y^2 - 9b Z^2)(Y^2 + 3b Z^2), Z3 = 8 Y^3 Z) has only TWO dependent levels of products, four products each:
//      {Y^2, Y Z, Z^2, X Y}   ->   {3b Z^2 * 8 Y^2,  Y Z * 8 Y^2,  (Y^2 - 9b Z^2)(Y^2 + 3b Z^2),  (Y^2 - 9b Z^2) * 2 X Y}
// against three for the Jacobian doubling above, and the accumulator never leaves the homogeneous form the additions of the
// Horner chain want (no conversion to Jacobian coordinates and back around every run of doublings, no special case for the
// identity).  Lane r of every quad computes product r of a level; a quad hands its four results round with DPP quad
// permutes — plain vector moves, no LDS round trip — and the three multiplications by small constants between the levels are one
// more per-lane step.  ~550 instructions per doubling on the critical path instead of ~800.
// Every lane must enter with the same point; Y is carried with limb magnitude 2 (the sum that ends a doubling is not
// normalised: the products of the next one have the room).
template <int K, class M>
__device__ __forceinline__ M msm_quad_bcast(const M& v) {
    M r;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(v.e.v) / sizeof(v.e.v[0])); i++)
        r.e.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.e.v[i], K * 0x55, 0xf, 0xf, false);
    return r;
}
template <class C>
__device__ __forceinline__ void msm_hom_dbl_quad(typename Field<C>::M1& X, Mag<C, 2, 2>& Y, typename Field<C>::M1& Z, int role) {
    using F = Field<C>;
    static_assert(C::A_IS_ZERO && C::REPR == REPR_U29_K256, "a = 0 on the 9 x 29 field");
    constexpr uint32_t b3 = 3 * C::B_SMALL;
    static_assert(3 * b3 < (1u << 12), "small-constant multiplier");
    // level 1:  Y^2 | Y Z | Z^2 | X Y
    const auto p1 = F::mul(F::sel(role == 2, Z, F::sel(role == 3, X, Y)), F::sel(role == 0 || role == 3, Y, Z));   // 2*2 = 4
    const auto yy = msm_quad_bcast<0>(p1), yz = msm_quad_bcast<1>(p1), zz = msm_quad_bcast<2>(p1), xy = msm_quad_bcast<3>(p1);
    // small constants:  3b Z^2 | 9b Z^2 | 8 Y^2 | (unused)
    const auto ps = F::template wrap<1, 1>(F::k_mul_small(F::sel(role < 2, zz, yy).e, role == 0 ? b3 : role == 1 ? 3 * b3 : 8u));
    const auto bzz3 = msm_quad_bcast<0>(ps), bzz9 = msm_quad_bcast<1>(ps), yy8 = msm_quad_bcast<2>(ps);
    const auto yy_m9 = F::sub(yy, bzz9);                           // 3
    const auto yy_p3 = F::add(yy, bzz3);                           // 2
    const auto xy2 = F::dbl(xy);                                   // 2
    // level 2:  3b Z^2 * 8 Y^2 | Y Z * 8 Y^2 | (Y^2 - 9b Z^2)(Y^2 + 3b Z^2) | (Y^2 - 9b Z^2) * 2 X Y
    const auto p2 = F::mul(F::sel(role == 0, bzz3, F::sel(role == 1, yz, yy_m9)),
                           F::sel(role < 2, yy8, F::sel(role == 2, yy_p3, xy2)));                                   // 3*2 = 6
    X = msm_quad_bcast<3>(p2);
    Z = msm_quad_bcast<1>(p2);
    Y = F::add(msm_quad_bcast<0>(p2), msm_quad_bcast<2>(p2));
}

// The complete addition of the chain (Group::add_a0's formulas) the same way: its twelve products are two dependent levels of
// three + three (the second with the subtractions of the first folded into its reduction, F::mul_sub) and one level of three
// two-product sums (F::mul2) — lanes 0..2 of every quad; ~1000 instructions on the critical path instead of ~2150.
// All coordinates enter and leave with magnitude 1.
template <class C>
__device__ __forceinline__ void msm_hom_add_quad(typename Field<C>::M1& X1, typename Field<C>::M1& Y1, typename Field<C>::M1& Z1,
                                                 const Proj<C>& q, int role) {
    using F = Field<C>;
    using G = Group<C>;
    static_assert(C::A_IS_ZERO && C::REPR == REPR_U29_K256, "a = 0 on the 9 x 29 field");
    constexpr uint32_t b3 = 3 * C::B_SMALL;
    const auto X2 = G::m(q.x), Y2 = G::m(q.y), Z2 = G::m(q.z);
    // level 1a:  X1 X2 | Y1 Y2 | Z1 Z2
    const auto pa = F::mul(F::sel(role == 1, Y1, F::sel(role == 2, Z1, X1)), F::sel(role == 1, Y2, F::sel(role == 2, Z2, X2)));
    const auto xx = msm_quad_bcast<0>(pa), yy = msm_quad_bcast<1>(pa), zz = msm_quad_bcast<2>(pa);
    // level 1b:  (X1 + Y1)(X2 + Y2) - (xx + yy) | (Y1 + Z1)(Y2 + Z2) - (yy + zz) | (X1 + Z1)(X2 + Z2) - (xx + zz)
    const auto pb = F::mul_sub(F::add(F::sel(role == 1, Y1, X1), F::sel(role == 0, Y1, Z1)),                        // 2*2 = 4
                               F::add(F::sel(role == 1, Y2, X2), F::sel(role == 0, Y2, Z2)),
                               F::add(F::sel(role == 1, yy, xx), F::sel(role == 0, yy, zz)));
    const auto xy = msm_quad_bcast<0>(pb), yz = msm_quad_bcast<1>(pb), xz = msm_quad_bcast<2>(pb);
    // small constants:  3b zz | 3b yz | 9b xx
    const auto ps = F::template wrap<1, 1>(F::k_mul_small(F::sel(role == 0, zz, F::sel(role == 1, yz, xx)).e, role == 2 ? 3 * b3 : b3));
    const auto bzz3 = msm_quad_bcast<0>(ps), byz3 = msm_quad_bcast<1>(ps), bxx9 = msm_quad_bcast<2>(ps);
    const auto xx3 = F::add(F::dbl(xx), xx);                       // 3
    const auto yy_m = F::norm(F::sub(yy, bzz3));                   // 3 -> 1
    const auto yy_p = F::add(yy, bzz3);                            // 2
    // level 2:  xy yy_m - 3b yz xz | yy_p yy_m + 9b xx xz | yz yy_p + 3 xx xy
    const auto p2 = F::mul2(F::sel(role == 0, xy, F::sel(role == 1, yy_p, yz)), F::sel(role == 2, yy_p, yy_m),      // 2*2 + 3*1 = 7
                            F::sel(role == 0, F::neg(byz3), F::sel(role == 1, bxx9, xx3)), F::sel(role == 2, xy, xz));
    X1 = msm_quad_bcast<0>(p2);
    Y1 = msm_quad_bcast<1>(p2);
    Z1 = msm_quad_bcast<2>(p2);
}

// workgroup-wide sum of one projective point per lane, result valid in lane 0 (block_sum, ecgpu_kernels.h) — for k256 with the
// tree's additions on quad lanes wherever a level has at most a quarter of the lanes busy (every level but the first): a level
// is then ~1,150 instructions (two LDS reads, msm_hom_add_quad, one write) instead of ~2,220, and these trees are pure
// latency — one workgroup per (part, window), a wave or less per SIMD.
template <class C>
__device__ __forceinline__ Proj<C> msm_block_sum(Proj<C> acc, uint32_t* lds, const Fe<C::NL>& b) {
    if constexpr (!(C::A_IS_ZERO && C::REPR == REPR_U29_K256)) {
        return block_sum<C>(acc, lds, b);
    } else {
        using G = Group<C>;
        constexpr int NL = C::NL;
        const int nt = (int)blockDim.x, t = (int)threadIdx.x;
        auto put = [&](int node, const Proj<C>& p) {
            uint32_t* d = lds + node * (3 * NL);
#pragma unroll
            for (int l = 0; l < NL; l++) { d[l] = p.x.v[l]; d[NL + l] = p.y.v[l]; d[2 * NL + l] = p.z.v[l]; }
        };
        auto get = [&](int node) {
            const uint32_t* o = lds + node * (3 * NL);
            Proj<C> q;
#pragma unroll
            for (int l = 0; l < NL; l++) { q.x.v[l] = o[l]; q.y.v[l] = o[NL + l]; q.z.v[l] = o[2 * NL + l]; }
            return q;
        };
        put(t, acc);
        __syncthreads();
        for (int s = nt / 2; s > 0; s >>= 1) {
            if (4 * s > nt) {                               // the first level: one lane per node
                if (t < s) acc = G::add(acc, get(t + s), b);
                __syncthreads();
                if (t < s) put(t, acc);
            } else {                                        // node i on the four lanes of quad i (whole quads are in or out)
                const int i = t >> 2;
                const bool in = i < s;
                if (in) {
                    const Proj<C> p = get(i), q = get(i + s);
                    auto X = G::m(p.x), Y = G::m(p.y), Z = G::m(p.z);
                    msm_hom_add_quad<C>(X, Y, Z, q, t & 3);
                    acc.x = X.e;
                    acc.y = Y.e;
                    acc.z = Z.e;
                }
                __syncthreads();
                if (in && (t & 3) == 0) put(i, acc);
            }
            __syncthreads();
        }
        return acc;                                         // lane 0: node 0 of the last level
    }
}

// ---- reduce ------------------------------------------------------------------------------------------------------------

// k * P for a small non-negative k (k < 2^31), two bits at a time from a table {P, 2P, 3P}.  The callers' lanes hold DIFFERENT k (the
// base weights of 64 consecutive segments), so in bit-by-bit double-and-add the wave executed the addition of almost every bit — some
// lane always has it set —: 15 doublings + 15 additions for a 15-bit weight.  With two-bit digits it is 2 doublings + ONE addition per
// digit whatever the lanes hold (a zero digit adds the identity: the formulas are complete), 17 doublings + 9 additions in all
// (round 6: k_msm_reduce_segments 0.236 -> 0.19 ms).
template <class C>
__device__ __forceinline__ Proj<C> small_mul(const Proj<C>& p, uint32_t k, const Fe<C::NL>& b) {
    using G = Group<C>;
    if (k == 0) return G::identity();
    if constexpr (C::NL > 9) {             // only k256 (nine limbs) keeps two waves per SIMD with three more live points; the other sets keep the bit-by-bit form
        Proj<C> acc = G::identity();
        const int top = 31 - __clz(k);
#pragma unroll 1
        for (int bit = top; bit >= 0; bit--) {
            acc = G::dbl(acc, b);
            if ((k >> bit) & 1) acc = G::add(acc, p, b);
        }
        return acc;
    }
    const Proj<C> p2 = G::dbl(p, b), p3 = G::add(p2, p, b);
    const int top = (31 - __clz(k)) | 1;                      // the upper bit of the leading two-bit digit
    Proj<C> acc = G::identity();
#pragma unroll 1
    for (int bit = top; bit >= 1; bit -= 2) {
        if (bit != top) acc = G::dbl(G::dbl(acc, b), b);
        const uint32_t d = (k >> (bit - 1)) & 3u;
        Proj<C> t = G::identity();
#pragma unroll
        for (int l = 0; l < C::NL; l++) {
            t.x.v[l] = d == 1 ? p.x.v[l] : d == 2 ? p2.x.v[l] : d == 3 ? p3.x.v[l] : t.x.v[l];
            t.y.v[l] = d == 1 ? p.y.v[l] : d == 2 ? p2.y.v[l] : d == 3 ? p3.y.v[l] : t.y.v[l];
            t.z.v[l] = d == 1 ? p.z.v[l] : d == 2 ? p2.z.v[l] : d == 3 ? p3.z.v[l] : t.z.v[l];
        }
        acc = bit == top ? t : G::add(acc, t, b);
    }
    return acc;
}

// segs[w][s] = sum_{j < seg} weight(s*seg + j) * buckets[w][s*seg + j],  weight(b) = (b >> shift_w) + 1 with
// shift_w = 0 except for the last window (sub-buckets, see msm_digit).  Running-sum trick: walking the
// segment downwards, `running` is added to `local` once per unit drop of the weight, and the weight of the
// lowest bucket multiplies the whole segment sum at the end.
template <class C>
__global__ void __launch_bounds__(64)
k_msm_reduce_segments(const uint32_t* __restrict__ buckets, size_t nb, int seg, size_t nseg, int nwin, int top_shift,
                      uint32_t* __restrict__ segs) {
    using G = Group<C>;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nseg * nwin) return;
    size_t w = gid / nseg, s = gid % nseg;
    Fe<C::NL> b = G::curve_b();
    Proj<C> running = G::identity(), local = G::identity();
    size_t base = s * seg;
    const int sh = (int)w == nwin - 1 ? top_shift : 0;
#pragma unroll 1
    for (int j = seg - 1; j >= 0; j--) {
        Proj<C> bk = load_proj<C>(buckets, w * nb + base + j);
        running = G::add(running, bk, b);
        if (j > 0 && ((base + j) >> sh) != ((base + j - 1) >> sh)) local = G::add(local, running, b);
    }
    uint32_t wmin = (uint32_t)(base >> sh) + 1;
    local = G::add(local, wmin == 1 ? running : small_mul<C>(running, wmin, b), b);
    store_proj<C>(segs, gid, local);
}

// The same with the bucket finish inside (round 5): the lane of a segment sums the stretches of its `seg` buckets itself — no
// k_msm_bucket_finish launch (two rounds of two waves per SIMD whose lanes wait for the slowest bucket of the wave: 0.16 ms at
// 2^21 terms for ~3 additions per lane) and no round trip of the bucket sums through HBM.  One wave per SIMD and the whole
// register file: nseg * nwin lanes are one wave per SIMD at 2^21 terms and two rounds of one at 2^24.  Buckets with
// MSM_BIG_PARTIALS stretches or more (degenerate scalar sets) were listed by k_msm_find_big and summed by k_msm_big_buckets
// before this kernel runs: they are read from `buckets`.
template <class C>
__global__ void __launch_bounds__(64, 1)
k_msm_finish_segments(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                      const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ pts,
                      const uint32_t* __restrict__ sorted, size_t n, size_t nb, int nwin, size_t chunk, size_t nchunks,
                      const uint32_t* __restrict__ buckets, int seg, size_t nseg, int top_shift, uint32_t* __restrict__ segs) {
    using G = Group<C>;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nseg * nwin) return;
    size_t w = gid / nseg, s = gid % nseg;
    const Fe<C::NL> b = G::curve_b();
    MsmPartialsHbm<C> src{const_cast<uint32_t*>(partials) + w * (nb + nchunks) * (4 * Field<C>::NS)};
    MsmPointsHbm<C> points{pts};
    Proj<C> running = G::identity(), local = G::identity();
    size_t base = s * seg;
    const int sh = (int)w == nwin - 1 ? top_shift : 0;
#pragma unroll 1
    for (int j = seg - 1; j >= 0; j--) {
        const size_t gb = w * nb + base + j;
        const uint32_t first = offsets[gb], cnt = counts[gb];
        Proj<C> bk;
        if (cnt != 0 && (first + cnt - 1) / (uint32_t)chunk - first / (uint32_t)chunk >= MSM_BIG_PARTIALS)
            bk = load_proj<C>(buckets, gb);
        else
            bk = msm_bucket_finish<C>((uint32_t)(base + j), first, cnt, (uint32_t)chunk, b, src, sorted + w * n, points);
        running = G::add(running, bk, b);
        if (j > 0 && ((base + j) >> sh) != ((base + j - 1) >> sh)) local = G::add(local, running, b);
    }
    uint32_t wmin = (uint32_t)(base >> sh) + 1;
    local = G::add(local, wmin == 1 ? running : small_mul<C>(running, wmin, b), b);
    store_proj<C>(segs, gid, local);
}

// parts[w][g] = sum of the segment sums segs[w][g * per .. (g + 1) * per): one workgroup per (g, w), a strided pass
// and an LDS tree.  With the default plan (4 buckets per segment, 256 segments per workgroup) a lane adds one segment:
// the depth is the 8 levels of the tree, where one workgroup per window used to walk 32 segments per lane first.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_msm_reduce_windows(const uint32_t* __restrict__ segs, size_t nseg, size_t per, uint32_t* __restrict__ parts) {
    using G = Group<C>;
    __shared__ uint32_t lds[BLOCK * 3 * C::NL];
    Fe<C::NL> b = G::curve_b();
    Proj<C> acc = G::identity();
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < nseg ? lo + per : nseg;
    for (size_t s = lo + threadIdx.x; s < hi; s += BLOCK)
        acc = G::add(acc, load_proj<C>(segs, (size_t)blockIdx.y * nseg + s), b);
    acc = msm_block_sum<C>(acc, lds, b);
    if (threadIdx.x == 0) store_proj<C>(parts, (size_t)blockIdx.y * gridDim.x + blockIdx.x, acc);
}

// wins[w] = sum over ranks r and workgroups g of parts[r][w][g] — the one place where the partial results of several
// GPUs meet (nranks = 1: this GPU's own).  One workgroup per window.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_msm_window_sums(const uint32_t* __restrict__ parts, int nranks, int nwin, int nparts, uint32_t* __restrict__ wins) {
    using G = Group<C>;
    __shared__ uint32_t lds[BLOCK * 3 * C::NL];
    Fe<C::NL> b = G::curve_b();
    Proj<C> acc = G::identity();
    const int items = nranks * nparts;
    for (int t = threadIdx.x; t < items; t += (int)blockDim.x) {
        const int r = t / nparts, g = t % nparts;
        acc = G::add(acc, load_proj<C>(parts, ((size_t)r * nwin + blockIdx.x) * nparts + g), b);
    }
    acc = msm_block_sum<C>(acc, lds, b);
    if (threadIdx.x == 0) store_proj<C>(wins, blockIdx.x, acc);
}

// ---- one Jacobian doubling spread over three lanes of a wave (the a = -3 sets; k256: msm_hom_dbl_quad below) ------------------
// The Horner chain below is ONE dependency chain of c * (nwin - 1) doublings (120 for 128-bit sub-scalars, 240 for 255-bit
// ones): a single lane issues one instruction every ~5 cycles, so the chain's time is its instruction count.  A doubling's
// seven or eight field multiplications are only three or four DEPENDENT levels:
//   a = -3 (dbl-2001-b)   {Z^2, Y^2, (Y + Z)^2}  ->  {X gamma, (X - delta)(X + delta), gamma^2}  ->  {alpha3^2}  ->  {alpha3 (4 beta - X3)}
// Lanes 0, 1, 2 of the wave each compute one product of a level (the same instruction stream on per-lane operands: plain
// SIMT), the three results are handed to every lane through the LDS crossbar (`__shfl`, 9-15 words each) and the cheap
// linear steps in between are done by all lanes alike, so that every lane holds the whole state again.  Three resp. four
// multiplication times per doubling instead of seven resp. eight.  Every lane must enter with the same point.
template <class M>
__device__ __forceinline__ M msm_lane_bcast(const M& v, int src) {
    M r;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(v.e.v) / sizeof(v.e.v[0])); i++) r.e.v[i] = (uint32_t)__shfl((int)v.e.v[i], src, 64);
    return r;
}
template <class C, class A, class B, class D>
__device__ __forceinline__ auto msm_sel3(int lane, const A& a, const B& b, const D& d) {
    using F = Field<C>;
    return F::sel(lane == 0, a, F::sel(lane == 1, b, d));
}

template <class C>
__device__ __forceinline__ Jac<C> msm_jac_dbl_lanes(const Jac<C>& p, int lane) {
    using G = Group<C>;
    using F = Field<C>;
    auto X = G::mj(p.x), Y = G::mj(p.y), Z = G::mj(p.z);
    Jac<C> o;
    static_assert(!C::A_IS_ZERO, "k256 takes the complete doublings on quad lanes (msm_hom_dbl_quad)");
    {
        const auto p1 = F::sqr(msm_sel3<C>(lane, Z, Y, F::add(Y, Z)));                              // delta | gamma | (Y + Z)^2
        const auto delta = msm_lane_bcast(p1, 0), gamma = msm_lane_bcast(p1, 1), yz = msm_lane_bcast(p1, 2);
        const auto p2 = F::mul(msm_sel3<C>(lane, X, F::sub(X, delta), gamma), msm_sel3<C>(lane, gamma, F::add(X, delta), gamma));
        const auto beta = msm_lane_bcast(p2, 0), alpha = msm_lane_bcast(p2, 1), gg = msm_lane_bcast(p2, 2);
        const auto alpha3 = F::add(F::dbl(alpha), alpha);                                           // 3
        const auto beta4 = F::dbl(F::dbl(beta));                                                    // 4
        const auto X3 = F::sqr_sub(alpha3, F::dbl(beta4));                             // 10 -> 1
        const auto gg8 = F::dbl(F::dbl(F::dbl(gg)));                                                // 8
        o.x = G::jstore(X3);
        o.y = G::jstore(F::mul_sub(alpha3, F::sub(beta4, X3), gg8));
        o.z = G::jstore(F::norm(F::sub(yz, F::add(gamma, delta))));
    }
    return o;
}

// out = sum_w 2^(c w) wins[w]   (Horner).  One wave; lanes 0..2 share the doublings (above), every lane carries the same
// accumulator, lane 0 stores.
// out_xy != nullptr: the result leaves the kernel as a wire record (affine x || y + identity flag, what k_normalize<C, NORM_WIRE>
// writes for one point) instead of a projective point in `out` — one launch and one load round trip less at the end of the chain.
template <class C>
__global__ void __launch_bounds__(64) k_msm_combine(const uint32_t* __restrict__ wins, int c, int nwin, uint32_t* __restrict__ out,
                                                    uint8_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
    using G = Group<C>;
    if (blockIdx.x != 0) return;
    const int lane = (int)threadIdx.x;
    // Everything here is wave-uniform, and left alone the compiler moves the whole doubling chain to the
    // scalar ALU (no 32x32+64 multiply-add there: 10x the instructions).  An opaque zero in a VGPR makes the
    // addresses, hence the data, formally divergent, which keeps the arithmetic on the vector ALU.
    uint32_t vzero = 0;
    asm volatile("" : "+v"(vzero));
    const uint32_t* vw = wins + vzero;
    Fe<C::NL> b = G::curve_b();
    Proj<C> acc = load_proj<C>(vw, nwin - 1);
    if constexpr (C::A_IS_ZERO && C::REPR == REPR_U29_K256) {
        // complete doublings and additions in the accumulator's own (homogeneous) form, the products of a level on the lanes of
        // a quad (msm_hom_dbl_quad, msm_hom_add_quad); every quad of the wave computes the same thing
        const int role = lane & 3;
        auto X = G::m(acc.x), Y = G::m(acc.y), Z = G::m(acc.z);
#if ECGPU_MSM_COMBINE_ROWS
        // round 5: the c doublings between two windows on the ROWS of the wave (ecgpu_rows.h: one limb per lane, the four products of
        // a level on the four rows; ~190 instructions per doubling instead of 569), the addition of a window's sum on quad lanes as before
        __shared__ uint32_t rows_lds[48];
        RowsDblK256 rd;
        rd.init();
#pragma unroll 1
        for (int w = nwin - 2; w >= 0; w--) {
            const Proj<C> q = load_proj<C>(vw, w);                      // in flight under the doublings
            uint32_t A, B, Q = 0;
            rd.enter(rows_lds, X.e, Y.e, Z.e, A, B);
#pragma unroll 1
            for (int s = 0; s < c; s++) Q = rd.step(A, B);
            Fe<C::NL> x2, y2, z2;
            rd.leave(Q, x2, y2, z2);
            X = Field<C>::template wrap<1, 1>(x2);
            Y = Field<C>::norm(Field<C>::template wrap<2, 2>(y2));
            Z = Field<C>::template wrap<1, 1>(z2);
            msm_hom_add_quad<C>(X, Y, Z, q, role);
        }
#else
#pragma unroll 1
        for (int w = nwin - 2; w >= 0; w--) {
            const Proj<C> q = load_proj<C>(vw, w);                      // in flight under the doublings
            auto Yw = Field<C>::template wrap<2, 2>(Y.e);
#pragma unroll 1
            for (int s = 0; s < c; s++) msm_hom_dbl_quad<C>(X, Yw, Z, role);
            Y = Field<C>::norm(Yw);
            msm_hom_add_quad<C>(X, Y, Z, q, role);
        }
#endif
        acc.x = X.e;
        acc.y = Y.e;
        acc.z = Z.e;
    } else {
        for (int w = nwin - 2; w >= 0; w--) {
            // c doublings in Jacobian coordinates (2M + 5S resp. 3M + 5S instead of the complete 6M + 2S + .. / 8M + 3S + ..):
            // (X : Y : Z) -> (X Z : Y Z^2 : Z) and back (X Z : Y : Z^3).  The identity has no Jacobian form here: skipped
            // (by every lane: the accumulator is the same in all of them).
            if (!G::is_identity(acc)) {
                Jac<C> j;
                {
                    auto X = G::m(acc.x), Y = G::m(acc.y), Z = G::m(acc.z);
                    j.x = Field<C>::mul(X, Z).e;
                    j.y = Field<C>::mul(Y, Field<C>::sqr(Z)).e;
                    j.z = acc.z;
                }
                if constexpr (GenericA<C>::value) {
                    for (int s = 0; s < c; s++) j = G::jac_dbl(j);                  // (any-a curves: the one-lane chain)
                } else {
#pragma unroll 1
                    for (int s = 0; s < c; s++) j = msm_jac_dbl_lanes<C>(j, lane);
                }
                acc = G::jac_to_proj(j);
            }
            acc = G::add(acc, load_proj<C>(vw, w), b);
        }
    }
    if (out_xy == nullptr) {
        if (lane == 0) store_proj<C>(out, 0, acc);
        return;
    }
    // `to_affine` (k256 projective.rs:64-75, primeorder projective.rs:74-86) on the one point; every lane computes, lane 0 stores
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    const bool ident = G::is_identity(acc);
    uint32_t wx[N], wy[N];
#pragma unroll
    for (int i = 0; i < N; i++) wx[i] = wy[i] = 0;
    if (!ident) {
        const typename F::M1 zinv = F::template inv<true>(G::m(acc.z));     // (an MSM's scalars are public: the variable-time steps)
        F::to_canonical(wx, F::mul(G::m(acc.x), zinv));
        F::to_canonical(wy, F::mul(G::m(acc.y), zinv));
    }
    if (lane == 0) {
        store_wire<C>(out_xy, wx);
        store_wire<C>(out_xy + WB, wy);
        if (out_inf) out_inf[0] = ident ? 1 : 0;
    }
}

// out[0 .. count) = the identity
template <class C>
__global__ void __launch_bounds__(BLOCK) k_store_identity(uint32_t* out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) store_proj<C>(out, i, Group<C>::identity());
}

// ---- plan ------------------------------------------------------------------------------------------------------------
// Window width from the term count, from sweeps on MI355X (tools/gpu_msm_sweep.py; plain: v11 kernels, GLV: r02b):
// plain, fastest c at 2^12 / 2^14 / 2^16 / 2^17 ... 2^19 / 2^20 / 2^21 ... = 9 / 11 / 12 / 13 / 14 / 16; GLV (2 n entries of
// 128 bits: c = 15 gives 9 full windows, c = 16 eight and a carry-only ninth): 13 up to 2^18 terms, 15 from 2^19.
// Below 2^17 the curve is flat: the parts that do not depend on n dominate whatever c is.
constexpr size_t MSM_GLV_MAX_TERMS = (size_t)13 << 17;      // 1.625 x 2^20 (msm_use_glv)
inline int msm_window_bits(size_t n, bool glv) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c;
    if (glv) c = lg <= 15 ? lg - 2 : (lg <= 18 ? 13 : 15);
    else if (lg <= 16) c = lg - 3;
    else if (lg <= 19) c = 13;
    else if (lg == 20) c = n < MSM_GLV_MAX_TERMS ? 14 : 16;     // (k256 takes the plain scalar from there on: 16, as for 2^21)
    else c = 16;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}

// GLV halves or the plain folded scalar?  k256 only, below MSM_GLV_MAX_TERMS terms (see MsmSplit).  Round 2 put the crossover above
// 2^21 terms (GLV / plain = 3.58 / 3.74 ms at 2^21, 6.46 / 6.16 at 2^22); since then the Horner chain — what the halves shorten —
// became three times cheaper (rows of a wave, round 5) while the halves still double prepare and sort, and the round-6 sweep with
// HEAD's kernels (tools/gpu_msm_crossover.py, profiles/r06/msm_glv_crossover_*.txt) reads, best GLV (c = 15) / best plain (c = 16):
// 1.467 / 1.547 ms at 2^20 terms, 2.528 / 2.468 at 2^21, 5.00 / 4.38 at 2^22 — the lines cross at about 1.6 x 2^20 terms.
// ECGPU_MSM_GLV = 0 / 1 forces it off / on, ECGPU_MSM_GLV_MAX_LOG2 moves the threshold to a power of two (tuning knobs of the tool
// build; results do not depend on them).
template <class C>
bool msm_use_glv(size_t n) {
    if (!MsmHasGlv<C>::value) return false;
    if (const char* e = knob("ECGPU_MSM_GLV")) {
        if (e[0] == '0') return false;
        if (e[0] == '1') return true;
    }
    if (const char* e = knob("ECGPU_MSM_GLV_MAX_LOG2")) {
        const int max_log2 = atoi(e);
        return max_log2 >= 0 && max_log2 < 40 && n <= ((size_t)1 << max_log2);
    }
    return n < MSM_GLV_MAX_TERMS;
}

// per-device launch facts (a context per GPU may plan concurrently: no unsynchronised function statics)
struct MsmDeviceFacts {
    std::atomic<size_t> wave_slots[64];
    std::atomic<bool> lds_attr[64];
};
inline MsmDeviceFacts& msm_device_facts() {
    static MsmDeviceFacts f{};
    return f;
}

// the window width an MSM of n terms gets (0 terms: the narrowest)
template <class C>
int msm_choose_window(size_t n) {
    return msm_window_bits(n ? n : 1, msm_use_glv<C>(n));
}

// glv: as msm_use_glv<C> decided for the term count the plan is made for (all GPUs of a sharded MSM: the same)
template <class C>
MsmPlan msm_plan(size_t n, int force_c, bool glv) {
    constexpr int N = C::N, NS = Field<C>::NS;
    MsmPlan p;
    p.glv = glv;
    p.npad = (n + 63) / 64 * 64;
    p.nsub = (glv ? 2 : 1) * p.npad;
    p.kbits = glv ? 128 : 32 * N - 1;
    const size_t ne = p.nsub;                               // entries per window the sort and the accumulation see
    p.c = force_c ? force_c : msm_window_bits(n ? n : 1, glv);
    p.nwin = signed_window_count(p.kbits, p.c);
    p.nb = (size_t)1 << (p.c - 1);
    p.seg = 4;                                              // buckets per running-sum lane: 4 ... 8 measured best for
                                                            // small MSMs (more lanes), neutral at 2^24 (tuning knob)
    if (const char* e = knob("ECGPU_MSM_SEG")) {
        int v = atoi(e);
        if (v >= 1 && v <= 1024 && (v & (v - 1)) == 0) p.seg = v;
    }
    if ((size_t)p.seg > p.nb) p.seg = (int)p.nb;
    p.nseg = p.nb / p.seg;
    p.nparts = (p.nseg + BLOCK - 1) / BLOCK;                // workgroups per window in the tree over the segment sums
    if (p.nparts > 32) p.nparts = 32;
    p.per_part = (p.nseg + p.nparts - 1) / p.nparts;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    int tile_log2 = 18;                                   // terms per counting-sort tile (tuning knob)
    if (const char* e = knob("ECGPU_MSM_TILE_LOG2")) {
        int v = atoi(e);
        if (v >= 12 && v <= 24) tile_log2 = v;
    }
    p.tile = (size_t)1 << tile_log2;
    p.ntiles = (ne + p.tile - 1) / p.tile;
    if (p.ntiles == 0) p.ntiles = 1;
    // Accumulation lanes: about three rounds of the wave slots the kernel can occupy (measured on MI355X: chunks
    // of ~500 entries beat one exactly-filling round of ~1400 by 2%, and anything that leaves slots empty loses
    // badly), but at least 32 entries per lane so that partial sums stay a small overhead.
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t>& slots = msm_device_facts().wave_slots[dev & 63];   // resident waves of k_msm_accumulate<C>
        size_t wave_slots = slots.load();
        if (!wave_slots) {
            int cus = 256, blocks = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_msm_accumulate<C>, 64, 0) != hipSuccess || blocks < 1)
                blocks = 8;
            wave_slots = (size_t)cus * blocks;
            slots.store(wave_slots);
        }
        size_t lanes = wave_slots * 64 * 3;
        size_t per_window = (lanes + p.nwin - 1) / p.nwin;
        p.chunk = (ne + per_window - 1) / per_window;
        if (p.chunk < 32) p.chunk = 32;
        // (no rounding to a multiple of four: the index stream starts anywhere inside a 16-byte quad, ecgpu_msm_chunk.h.  Round 4
        // rounded up — 57 -> 60 entries at 2^21 GLV terms, and the third round of waves was 16 % empty)
        if (const char* e = knob("ECGPU_MSM_CHUNK")) {
            long v = atol(e);
            if (v >= 1 && v <= (1L << 30)) p.chunk = (size_t)v;
        }
        p.nchunks = (ne + p.chunk - 1) / p.chunk;
        if (p.nchunks == 0) p.nchunks = 1;
    }
    {
        bool two = ne >= ((size_t)1 << 17);      // measured: equal at 2^16, 12 % faster at 2^18, 14 % at 2^24
        if (const char* e = knob("ECGPU_MSM_SORT2")) two = atoi(e) != 0;
        if (two && p.c - 1 > MSM_SORT2_BITS_A) {
            p.sort_bits_b = p.c - 1 - MSM_SORT2_BITS_A;
            // packed form: as many low bucket bits as fit beside the index and the sign (fewer low bits = more level-A keys)
            int idx_bits = 1;
            while (((size_t)1 << idx_bits) < ne) idx_bits++;
            int bb = p.sort_bits_b;
            if (bb > 31 - idx_bits) bb = 31 - idx_bits;
            if (bb > MSM_SORTP_MAX_BITS_B) bb = MSM_SORTP_MAX_BITS_B;
            bool packed = bb >= 1 && p.c - 1 - bb <= MSM_SORTP_MAX_BITS_A;
            // k_msm_prepare keeps the level-A histogram of every window in LDS (nwin * npart words) and is launched without the
            // large-LDS attribute: with 9 level-A bits (more than 2^24 entries per window) the 33 windows of p521 would need 67.6 KB
            // — such a plan keeps the unpacked form with its 8 level-A bits
            packed = packed && (size_t)p.nwin * (p.nb >> bb) * 4 <= (size_t)64 * 1024;
            if (const char* e = knob("ECGPU_MSM_SORT_PACKED")) packed = packed && atoi(e) != 0;   // 0: the round-3 kernels (A/B runs)
            if (packed) {
                p.sort_packed = true;
                p.idx_bits = idx_bits;
                p.sort_bits_b = bb;
            }
            p.npart = p.nb >> p.sort_bits_b;
            p.ntiles2 = (ne + MSM_SORT2_TILE - 1) / MSM_SORT2_TILE;
            p.off_tmpidx = o;  o = align(o + (size_t)p.nwin * ne * 4);
            if (!p.sort_packed) { p.off_tmpkey = o;  o = align(o + (size_t)p.nwin * ne * 2); }
            p.off_count_a = o;  o = align(o + (size_t)p.nwin * p.npart * 4);
            p.off_offset_a = o; o = align(o + ((size_t)p.nwin * p.npart + p.nwin) * 4);    // + nwin flags (big_any)
            p.off_cursor = o;   o = align(o + (size_t)p.nwin * p.nb * 4);
        }
    }
    p.off_points = o;  o = align(o + ne * 2 * N * 4);
    p.off_digits = o;  o = align(o + (size_t)p.nwin * ne * 2);
    p.off_vmask = o;   o = align(o + (size_t)p.nwin * (ne / 64) * 8);
    p.off_tilehist = o; o = align(o + (size_t)p.nwin * p.ntiles * p.nb * 4);
    p.off_sorted = o;  o = align(o + (size_t)p.nwin * ne * 4);
    p.off_count = o;   o = align(o + (size_t)p.nwin * p.nb * 4);
    p.off_offset = o;  o = align(o + (size_t)p.nwin * p.nb * 4);
    p.off_partials = o; o = align(o + (size_t)p.nwin * (p.nb + p.nchunks) * 4 * NS * 4);
    p.off_buckets = o; o = align(o + (size_t)p.nwin * p.nb * 3 * NS * 4);
    p.max_big = (size_t)p.nwin * (p.nchunks / (MSM_BIG_PARTIALS - 1) + 1);   // a big bucket covers >= 31 whole chunks
    p.off_biglist = o; o = align(o + (p.max_big + 1) * 4);
    p.off_segs = o;    o = align(o + (size_t)p.nwin * p.nseg * 3 * NS * 4);
    p.parts_bytes = (size_t)p.nwin * p.nparts * 3 * NS * 4;
    p.off_parts = o;   o = align(o + p.parts_bytes);
    p.off_wins = o;    o = align(o + (size_t)p.nwin * 3 * NS * 4);
    p.workspace_bytes = o + 256;
    return p;
}

// Everything between the accumulation and the per-window partial sums: bucket finish, running sums over segments of buckets,
// the tree over the segment sums.  These kernels hold a few thousand waves at most and each lane walks a chain of dependent
// point operations: their time is latency, not throughput.
template <class C>
void launch_msm_tail(const MsmPlan& p, hipStream_t stream, uint8_t* ws, uint32_t* parts) {
    const size_t ne = p.nsub;
    uint32_t* pts = (uint32_t*)(ws + p.off_points);
    uint32_t* sorted = (uint32_t*)(ws + p.off_sorted);
    uint32_t* counts = (uint32_t*)(ws + p.off_count);
    uint32_t* offsets = (uint32_t*)(ws + p.off_offset);
    uint32_t* partials = (uint32_t*)(ws + p.off_partials);
    uint32_t* buckets = (uint32_t*)(ws + p.off_buckets);
    uint32_t* big_list = (uint32_t*)(ws + p.off_biglist);
    uint32_t* segs = (uint32_t*)(ws + p.off_segs);
    const size_t nbk = p.nb * p.nwin;
    size_t nsg = p.nseg * p.nwin;
    if (msm_fused_tail<C>(p)) {
        // (k_msm_find_big ran before the accumulation: launch_msm_parts)
        hipLaunchKernelGGL(k_msm_big_buckets<C>, dim3(MSM_BIG_GRID), dim3(BLOCK), 0, stream, (const uint32_t*)partials,
                           (const uint32_t*)counts, (const uint32_t*)offsets, (const uint32_t*)pts, (const uint32_t*)sorted, ne, p.nb,
                           p.chunk, p.nchunks, buckets, (const uint32_t*)big_list, (uint32_t)p.max_big);
        hipLaunchKernelGGL((k_msm_finish_segments<C>), dim3((unsigned)((nsg + 63) / 64)), dim3(64), 0, stream,
                           (const uint32_t*)partials, (const uint32_t*)counts, (const uint32_t*)offsets, (const uint32_t*)pts,
                           (const uint32_t*)sorted, ne, p.nb, p.nwin, p.chunk, p.nchunks, (const uint32_t*)buckets, p.seg, p.nseg,
                           msm_top_shift(p.kbits, p.c), segs);
    } else {
        // (big_list[0] = 0 was written by the first lane of k_msm_accumulate)
        hipLaunchKernelGGL((k_msm_bucket_finish<C>), dim3((unsigned)((nbk + 63) / 64)), dim3(64), 0, stream,
                           (const uint32_t*)partials, (const uint32_t*)counts, (const uint32_t*)offsets, (const uint32_t*)pts,
                           (const uint32_t*)sorted, ne, p.nb, p.nwin, p.chunk, p.nchunks, buckets, big_list, (uint32_t)p.max_big);
        hipLaunchKernelGGL(k_msm_big_buckets<C>, dim3(MSM_BIG_GRID), dim3(BLOCK), 0, stream, (const uint32_t*)partials,
                           (const uint32_t*)counts, (const uint32_t*)offsets, (const uint32_t*)pts, (const uint32_t*)sorted, ne, p.nb,
                           p.chunk, p.nchunks, buckets, (const uint32_t*)big_list, (uint32_t)p.max_big);
        hipLaunchKernelGGL((k_msm_reduce_segments<C>), dim3((unsigned)((nsg + 63) / 64)), dim3(64), 0, stream,
                           (const uint32_t*)buckets, p.nb, p.seg, p.nseg, p.nwin, msm_top_shift(p.kbits, p.c), segs);
    }
    if (p.detail[1]) (void)hipEventRecord(p.detail[1], stream);
    hipLaunchKernelGGL((k_msm_reduce_windows<C>), dim3((unsigned)p.nparts, (unsigned)p.nwin), dim3(BLOCK), 0, stream,
                       (const uint32_t*)segs, p.nseg, p.per_part, parts);
    if (p.detail[2]) (void)hipEventRecord(p.detail[2], stream);
}

// First half of the pipeline: everything up to the per-window partial sums parts[nwin][nparts] (projective, internal
// form, plan.parts_bytes bytes) — what a GPU contributes to an MSM whose terms are spread over several GPUs.
template <class C>
void launch_msm_parts(const MsmPlan& p, hipStream_t stream, const uint8_t* d_scalars, const uint8_t* d_xy,
                      const uint8_t* d_inf, size_t n, void* workspace, uint32_t* parts, int* d_status, hipEvent_t ev_sorted,
                      hipEvent_t ev_accumulated) {
    const size_t nparts_total = (size_t)p.nwin * p.nparts;
    if (n == 0) {
        hipLaunchKernelGGL(k_store_identity<C>, dim3((unsigned)((nparts_total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, parts,
                           nparts_total);
        if (ev_sorted) (void)hipEventRecord(ev_sorted, stream);
        if (ev_accumulated) (void)hipEventRecord(ev_accumulated, stream);
        return;
    }
    const size_t ne = p.nsub;
    uint8_t* ws = (uint8_t*)workspace;
    uint32_t* pts = (uint32_t*)(ws + p.off_points);
    uint16_t* digits = (uint16_t*)(ws + p.off_digits);
    unsigned long long* vmask = (unsigned long long*)(ws + p.off_vmask);
    uint32_t* tile_hist = (uint32_t*)(ws + p.off_tilehist);
    uint32_t* sorted = (uint32_t*)(ws + p.off_sorted);
    uint32_t* counts = (uint32_t*)(ws + p.off_count);
    uint32_t* offsets = (uint32_t*)(ws + p.off_offset);
    uint32_t* partials = (uint32_t*)(ws + p.off_partials);
    // k_msm_prepare: each workgroup takes `reps` groups of BLOCK terms, so that its LDS histogram is flushed once for all of
    // them (one global atomic per counter and workgroup); at least ~1024 workgroups — four per CU, one resident at a time —
    // stay in the launch (measured, round 4: 2048 -> 1024 workgroups −3…5 % of the kernel; the flush order does not matter)
    int reps = 1;
    if (p.sort_bits_b) {
        while (reps < 32 && (n + (size_t)BLOCK * reps * 2 - 1) / ((size_t)BLOCK * reps * 2) >= 1024) reps *= 2;
    }
    unsigned g = (unsigned)((n + (size_t)BLOCK * reps - 1) / ((size_t)BLOCK * reps));
    const size_t lds_bytes = p.nb * 4;
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<bool>& done = msm_device_facts().lds_attr[dev & 63];
        if (!done.load()) {   // the window histogram may take 128 KiB of the 160 KiB LDS (an attribute of the function on this device)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done.store(true);
        }
    }
    uint32_t* counts_a0 = p.sort_bits_b ? (uint32_t*)(ws + p.off_count_a) : nullptr;
    const size_t prep_lds = p.sort_bits_b ? (size_t)p.nwin * p.npart * 4 : 0;
    if (p.sort_bits_b) {
        (void)hipMemsetAsync(counts_a0, 0, (size_t)p.nwin * p.npart * 4, stream);
        if (!p.sort_packed) (void)hipMemsetAsync(counts, 0, (size_t)p.nwin * p.nb * 4, stream);
    }
    if constexpr (MsmHasGlv<C>::value) {
        if (p.glv)
            hipLaunchKernelGGL((k_msm_prepare<C, true>), dim3(g), dim3(BLOCK), prep_lds, stream, d_scalars, d_xy, d_inf, n, p.npad, p.c,
                               p.nwin, pts, digits, vmask, d_status, counts_a0, p.sort_bits_b, (int)p.npart, reps);
    }
    if (!p.glv)
        hipLaunchKernelGGL((k_msm_prepare<C, false>), dim3(g), dim3(BLOCK), prep_lds, stream, d_scalars, d_xy, d_inf, n, p.npad, p.c,
                           p.nwin, pts, digits, vmask, d_status, counts_a0, p.sort_bits_b, (int)p.npart, reps);
    if (p.detail[0]) (void)hipEventRecord(p.detail[0], stream);
    if (p.sort_packed) {
        uint32_t* tmp = (uint32_t*)(ws + p.off_tmpidx);
        uint32_t* counts_a = (uint32_t*)(ws + p.off_count_a);
        uint32_t* offsets_a = (uint32_t*)(ws + p.off_offset_a);
        uint32_t* cursor = (uint32_t*)(ws + p.off_cursor);
        const uint32_t big = msm_sortb_big_limit(ne, p.npart);
        uint32_t* big_any = offsets_a + (size_t)p.nwin * p.npart;       // nwin flags behind the partition offsets (plan: + nwin words)
        hipLaunchKernelGGL(k_msm_scan, dim3(p.nwin), dim3(1024), 0, stream, (const uint32_t*)counts_a, offsets_a, p.npart, big, big_any,
                           cursor);                                     // (offsets and the scatter's cursors in one launch)
        hipLaunchKernelGGL(k_msm_sort_a, dim3((unsigned)p.ntiles2, (unsigned)p.nwin), dim3(1024), 0, stream, (const uint16_t*)digits,
                           (const unsigned long long*)vmask, ne, p.sort_bits_b, p.idx_bits, (uint32_t)p.npart, cursor, tmp);
        // level B: one workgroup per (partition, window); small partitions (small MSMs) get 256 lanes
        unsigned tb = ne / p.npart >= 4096 ? 1024u : 256u;
        if (const char* e = knob("ECGPU_MSM_SORTB_T")) {            // tuning knob: lanes per level-B workgroup
            const int v = atoi(e);
            if (v == 256 || v == 512 || v == 1024) tb = (unsigned)v;
        }
        hipLaunchKernelGGL(k_msm_sort_b, dim3((unsigned)p.npart, (unsigned)p.nwin), dim3(tb), 0, stream, (const uint32_t*)tmp, ne,
                           p.sort_bits_b, p.idx_bits, (uint32_t)p.npart, big, (const uint32_t*)offsets_a, (const uint32_t*)counts_a,
                           counts, offsets, sorted);
        // partitions above `big` entries (degenerate scalar sets): count / offsets / scatter over the whole chip, tiles of 8192
        // positions dealt out to at most 256 workgroups per window; for random scalars big_any[] is all zero and every
        // workgroup of the three launches exits on its first load
        const unsigned gb = (unsigned)(p.ntiles2 < 256 ? p.ntiles2 : 256);
        uint32_t* cursor_b = (uint32_t*)(ws + p.off_cursor);        // (level A's cursors are spent by now; nwin x nb words)
        hipLaunchKernelGGL((k_msm_sort_b_big<true>), dim3(gb, (unsigned)p.nwin), dim3(1024), 0, stream, (const uint32_t*)tmp, ne,
                           p.sort_bits_b, p.idx_bits, (uint32_t)p.npart, big, (const uint32_t*)offsets_a, (const uint32_t*)counts_a,
                           (const uint32_t*)big_any, counts, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_msm_sort_b_big_offsets, dim3(gb < p.npart ? gb : (unsigned)p.npart, (unsigned)p.nwin), dim3(256), 0, stream,
                           p.sort_bits_b, (uint32_t)p.npart, big, (const uint32_t*)offsets_a, (const uint32_t*)counts_a,
                           (const uint32_t*)big_any, (const uint32_t*)counts, offsets, cursor_b);
        hipLaunchKernelGGL((k_msm_sort_b_big<false>), dim3(gb, (unsigned)p.nwin), dim3(1024), 0, stream, (const uint32_t*)tmp, ne,
                           p.sort_bits_b, p.idx_bits, (uint32_t)p.npart, big, (const uint32_t*)offsets_a, (const uint32_t*)counts_a,
                           (const uint32_t*)big_any, cursor_b, sorted);
    } else if (p.sort_bits_b) {
        uint32_t* tmp_idx = (uint32_t*)(ws + p.off_tmpidx);
        uint16_t* tmp_key = (uint16_t*)(ws + p.off_tmpkey);
        uint32_t* counts_a = (uint32_t*)(ws + p.off_count_a);
        uint32_t* offsets_a = (uint32_t*)(ws + p.off_offset_a);
        uint32_t* cursor = (uint32_t*)(ws + p.off_cursor);
        const dim3 grid2((unsigned)p.ntiles2, (unsigned)p.nwin);
        MsmSort2Src sa{digits, vmask, nullptr, nullptr, nullptr, 0};
        // (the level-A histogram counts_a was taken by k_msm_prepare)
        hipLaunchKernelGGL(k_msm_scan, dim3(p.nwin), dim3(1024), 0, stream, (const uint32_t*)counts_a, offsets_a, p.npart);
        (void)hipMemcpyAsync(cursor, offsets_a, (size_t)p.nwin * p.npart * 4, hipMemcpyDeviceToDevice, stream);
        hipLaunchKernelGGL((k_msm_sort2<false, false>), grid2, dim3(1024), 0, stream, sa, ne, p.sort_bits_b, p.npart, cursor,
                           tmp_idx, tmp_key);
        MsmSort2Src sb{tmp_key, nullptr, tmp_idx, offsets_a, counts_a, p.npart};
        hipLaunchKernelGGL((k_msm_sort2<true, true>), grid2, dim3(1024), 0, stream, sb, ne, p.sort_bits_b, p.nb, counts,
                           (uint32_t*)nullptr, (uint16_t*)nullptr);
        hipLaunchKernelGGL(k_msm_scan, dim3(p.nwin), dim3(1024), 0, stream, (const uint32_t*)counts, offsets, p.nb);
        (void)hipMemcpyAsync(cursor, offsets, (size_t)p.nwin * p.nb * 4, hipMemcpyDeviceToDevice, stream);
        hipLaunchKernelGGL((k_msm_sort2<true, false>), grid2, dim3(1024), 0, stream, sb, ne, p.sort_bits_b, p.nb, cursor,
                           sorted, (uint16_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_msm_hist, dim3((unsigned)p.ntiles, (unsigned)p.nwin), dim3(1024), lds_bytes, stream,
                           (const uint16_t*)digits, (const unsigned long long*)vmask, ne, p.tile, p.nb, tile_hist);
        size_t nbk0 = p.nb * p.nwin;
        hipLaunchKernelGGL(k_msm_tile_scan, dim3((unsigned)((nbk0 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, tile_hist,
                           p.ntiles, p.nb, p.nwin, counts);
        hipLaunchKernelGGL(k_msm_scan, dim3(p.nwin), dim3(1024), 0, stream, (const uint32_t*)counts, offsets, p.nb);
        hipLaunchKernelGGL(k_msm_scatter, dim3((unsigned)p.ntiles, (unsigned)p.nwin), dim3(1024), lds_bytes, stream,
                           (const uint16_t*)digits, (const unsigned long long*)vmask, ne, p.tile, p.nb,
                           (const uint32_t*)tile_hist, (const uint32_t*)offsets, sorted);
    }
    if (msm_fused_tail<C>(p)) {                         // the buckets k_msm_big_buckets will take, listed while nothing waits for it
        uint32_t* big_list = (uint32_t*)(ws + p.off_biglist);
        const size_t nbk = p.nb * p.nwin;
        (void)hipMemsetAsync(big_list, 0, 4, stream);
        hipLaunchKernelGGL(k_msm_find_big, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)counts,
                           (const uint32_t*)offsets, nbk, (uint32_t)p.chunk, big_list, (uint32_t)p.max_big);
    }
    if (ev_sorted) (void)hipEventRecord(ev_sorted, stream);
    size_t nlanes = p.nchunks * p.nwin;
    hipLaunchKernelGGL(k_msm_accumulate<C>, dim3((unsigned)((nlanes + 63) / 64)), dim3(64), 0, stream,
                       (const uint32_t*)pts, (const uint32_t*)sorted, (const uint32_t*)counts,
                       (const uint32_t*)offsets, ne, p.nb, p.nwin, p.chunk, p.nchunks, partials,
                       msm_fused_tail<C>(p) ? (uint32_t*)nullptr : (uint32_t*)(ws + p.off_biglist));
    if (ev_accumulated) (void)hipEventRecord(ev_accumulated, stream);       // the accumulation kernel alone (round 4 recorded this after the bucket finish)
    launch_msm_tail<C>(p, stream, ws, parts);
}

// Second half: the window sums over `nranks` sets of partial sums (laid out [rank][nwin][nparts]) and the Horner chain
// over the windows; the result lands in out[0] (projective, internal form) or, if out_xy is given, in out_xy / out_inf as a
// wire record.  `wins` is nwin points of scratch.
template <class C>
void launch_msm_finish(const MsmPlan& p, hipStream_t stream, const uint32_t* parts_all, int nranks, uint32_t* wins, uint32_t* out,
                       uint8_t* out_xy, uint8_t* out_inf) {
    // the tree of k_msm_window_sums is as wide as the parts of all ranks need, not wider (16 parts: 4 levels, not 8)
    int items = nranks * (int)p.nparts, block = 64;
    while (block < items && block < BLOCK) block *= 2;
    hipLaunchKernelGGL(k_msm_window_sums<C>, dim3((unsigned)p.nwin), dim3(block), 0, stream, parts_all, nranks, p.nwin, (int)p.nparts, wins);
    hipLaunchKernelGGL(k_msm_combine<C>, dim3(1), dim3(64), 0, stream, (const uint32_t*)wins, p.c, p.nwin, out, out_xy, out_inf);
}

// The whole pipeline on one GPU.
template <class C>
void launch_msm(const MsmPlan& p, hipStream_t stream, const uint8_t* d_scalars, const uint8_t* d_xy,
                const uint8_t* d_inf, size_t n, void* workspace, uint32_t* out, int* d_status, hipEvent_t ev_sorted,
                hipEvent_t ev_accumulated, uint8_t* out_xy, uint8_t* out_inf) {
    uint8_t* ws = (uint8_t*)workspace;
    uint32_t* parts = (uint32_t*)(ws + p.off_parts);
    launch_msm_parts<C>(p, stream, d_scalars, d_xy, d_inf, n, workspace, parts, d_status, ev_sorted, ev_accumulated);
    launch_msm_finish<C>(p, stream, parts, 1, (uint32_t*)(ws + p.off_wins), out, out_xy, out_inf);
}

}  // namespace ecgpu
