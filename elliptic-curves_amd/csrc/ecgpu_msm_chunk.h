// ecgpu_msm_chunk.h — the per-lane body of the Pippenger bucket accumulation (host + device; tests/hostcheck
// runs exactly this code on the CPU).
//
// After the counting sort every window holds one sorted run of (sign, term index) entries in which each bucket
// is a contiguous stretch.  A lane does NOT own a bucket: it owns `chunk` consecutive entries of the run,
// wherever the bucket boundaries fall, so every lane performs the same number of additions no matter how
// the scalars are distributed (a bucket that received all 2^24 terms is spread over thousands of lanes).
// Whenever the lane leaves a bucket — or its chunk ends inside one — it writes the partial sum of the stretch to
//
//      partial[b + q]          b = bucket, q = chunk index within the window
//
// Buckets and chunks both advance monotonically along the run, so distinct (b, q) stretches get distinct
// slots, nb + nchunks slots suffice, and the stretches of one bucket occupy consecutive slots
// b + q_first .. b + q_last, which msm_bucket_finish adds up.
//
// A stretch is summed in XYZZ coordinates with the incomplete mixed addition (8M + 2S against 11M for the complete
// one), which is wrong exactly when some step adds +-(the running sum) — duplicates of a point in one bucket, or
// P + Q and (P + Q) handed in as a third term.  Every such step has U2 - X1 = 0 and multiplies ZZ by zero, and ZZ,
// a product of the steps' (U2 - X1)^2, cannot become zero in any other way: ZZ == 0 at the end of the stretch is the
// exact test, and a stretch that fails it is summed again with the complete formulas.  Both happen where the partial
// sums are consumed (msm_stretch_sum, called by the bucket finish), not in the accumulation loop: there, a lane that
// leaves a bucket makes its whole wave walk the exit path, so that path is four stores and nothing else.
#pragma once

#include "ecgpu_point.h"

namespace ecgpu {

// The sorted (sign, index) entries of a lane's chunk, fetched four at a time: one 16-byte load per four additions
// instead of a 4-byte one per addition (a quarter of the requests, and a lane's 128-byte line is asked for 8 times
// instead of 32) — and one quad AHEAD: the quad after the current one is requested when the current one is entered, so
// the load has four additions to complete in (round 4 requested a quad with the last entry of its predecessor and needed
// its first entry — the address of the next point — in the same step: every fourth step the wave sat out a load).
// `run` must be 16-byte aligned and readable up to the second multiple of four past the last entry taken (the device
// workspace continues behind the runs; entries past the chunk are requested, never used).
struct MsmIndexStream {
    const uint32_t* run;
    uint32_t b0, b1, b2, b3;                         // the current quad, rotating
    uint32_t n0, n1, n2, n3;                         // the quad after it
    ECGPU_HD void fetch_next(uint32_t pos4) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint4 v = *reinterpret_cast<const uint4*>(run + pos4);
        n0 = v.x; n1 = v.y; n2 = v.z; n3 = v.w;
#else
        n0 = run[pos4]; n1 = run[pos4 + 1]; n2 = run[pos4 + 2]; n3 = run[pos4 + 3];
#endif
    }
    ECGPU_HD void rotate() { b0 = b1; b1 = b2; b2 = b3; }
    // positions first, first + 1, ... are then handed out by take()
    ECGPU_HD void start(const uint32_t* r, uint32_t first) {
        run = r;
        fetch_next(first & ~3u);
        b0 = n0; b1 = n1; b2 = n2; b3 = n3;
        fetch_next((first & ~3u) + 4);
        for (uint32_t s = first & 3u; s != 0; s--) rotate();
    }
    // the entry at `pos` (calls must come in position order)
    ECGPU_HD uint32_t take(uint32_t pos) {
        const uint32_t e = b0;
        if (((pos + 1) & 3u) == 0) {
            b0 = n0; b1 = n1; b2 = n2; b3 = n3;
            fetch_next(pos + 5);
        } else {
            rotate();
        }
        return e;
    }
};

// The limbs of `a` are computed, and every memory operation that follows in the source is issued, on their own side of this
// point.  The accumulation loop needs it: the one wait of a step is the one for the point prefetched a whole addition ago, and it
// has to come BEFORE the step's own stores and loads are issued — gfx9 counts loads and stores in one counter (vmcnt) that the
// compiler can only wait to zero once both kinds are in flight, so a wait placed after them sits out their whole latency.
template <class C>
ECGPU_HD void msm_pin_before_memory_ops(Affine<C>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < C::NL; i++) asm volatile("" : "+v"(a.x.v[i]), "+v"(a.y.v[i]) : : "memory");
#else
    (void)a;
#endif
}

// Points: void load(PackedPoint<2N>&, uint32_t term) const.   Sink: void put(size_t slot, const Xyzz<C>&).
// `ow` = bucket start offsets of this window (nb entries), `total` = length of the window's run.
template <class C, class Points, class Sink>
ECGPU_HD void msm_chunk_accumulate(const uint32_t* __restrict__ run, const uint32_t* __restrict__ ow, uint32_t total,
                                   uint32_t nb, uint32_t chunk, uint32_t q, const Fe<C::NL>& curve_b,
                                   const Points& points, Sink& sink) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    const uint32_t start = q * chunk;
    if (start >= total) return;
    const uint32_t end = total - start < chunk ? total : start + chunk;
    // bucket containing `start`: the last b with ow[b] <= start (an empty bucket never qualifies as "last")
    uint32_t lo = 0, hi = nb;                       // invariant: ow[lo] <= start, (hi == nb or ow[hi] > start)
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ow[mid] <= start) lo = mid;
        else hi = mid;
    }
    uint32_t b = lo;
    uint32_t bend = b + 1 < nb ? ow[b + 1] : total;
    uint32_t bnext = b + 2 < nb ? ow[b + 2] : total;  // the end of the bucket after this one: asked for one bucket ahead (below)
    Xyzz<C> acc;
    acc.x = acc.y = acc.zz = acc.zzz = F::one().e;
    bool fresh = true;                               // no term of the current stretch taken yet
    bool pending = false;                            // the stretch that ended with the previous entry is still to be written
    uint32_t pslot = 0;
    PackedPoint<2 * N> pw;
    MsmIndexStream idx;
    idx.start(run, start);
    uint32_t e = idx.take(start);
    points.load(pw, e & 0x7FFFFFFFu);
#pragma unroll 1
    for (uint32_t pos = start; pos < end;) {
        Affine<C> cur;
        cur.x = F::unpack(pw.w).e;
        cur.y = F::unpack(pw.w + N).e;
        const bool neg = (e >> 31) != 0;
        pos++;
        msm_pin_before_memory_ops(cur);
        // Every memory operation of a step is issued HERE, at its head, and is a whole addition old when the next head waits
        // for the prefetched point: the partial sum of a stretch that ended with the previous entry (a lane that leaves a bucket
        // makes its whole wave walk this path, so it must not wait on anything: round 4 stored at the end of the step and then
        // read ow[b + 1] for the new bucket, i.e. drained the stores AND a dependent load with the wave stalled — at 128
        // entries per bucket, the 2^21-term share of an 8-GPU run, four steps of five have a leaving lane in the wave), the
        // bucket end after next, the next point.
        if (pending) {
            sink.put((size_t)pslot, acc);            // as it is: the exactness test and the conversion happen in the finish
            bnext = b + 2 < nb ? ow[b + 2] : total;
            pending = false;
        }
        {   // the next point under the current addition; unconditionally (after the last entry: the same point again, unused) so
            // that the registers of the point just unpacked can take it without a copy at the loop's back edge
            const uint32_t en = idx.take(pos);
            e = pos < end ? en : e;
            points.load(pw, e & 0x7FFFFFFFu);
        }
        if (fresh) {
            acc = G::xyzz_from_affine(cur, neg);
            fresh = false;
        } else {
            acc = G::xyzz_madd(acc, cur, neg);
        }
        if (pos == bend || pos == end) {             // leaving the bucket, or the chunk ends inside it
            pslot = b + q;
            pending = true;
            fresh = true;
            if (pos == bend && pos < end) {
                b++;
                bend = bnext;                        // known since the head of the step that entered the bucket left now
                if (bend == pos) {                   // empty buckets (rare): walk on with loads that are waited for
                    do {
                        b++;
                        bend = b + 1 < nb ? ow[b + 1] : total;
                    } while (bend == pos);           // pos < end <= total terminates this
                }
            }
        }
    }
    if (pending) sink.put((size_t)pslot, acc);
}

// The sum of one stretch from its stored XYZZ partial: converted if the exactness test passes, otherwise recomputed
// from the entries run[lo, hi) with the complete formulas.
template <class C, class Points>
ECGPU_HD Proj<C> msm_stretch_sum(const Xyzz<C>& part, const uint32_t* __restrict__ run, uint32_t lo, uint32_t hi,
                                 const Fe<C::NL>& curve_b, const Points& points) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    if (!F::is_zero(G::mj(part.zz))) return G::xyzz_to_proj(part);
    Proj<C> sum = G::identity();
#pragma unroll 1
    for (uint32_t r = lo; r < hi; r++) {
        const uint32_t er = run[r];
        PackedPoint<2 * N> pr;
        points.load(pr, er & 0x7FFFFFFFu);
        Affine<C> a;
        a.x = F::unpack(pr.w).e;
        a.y = F::unpack(pr.w + N).e;
        sum = G::add_mixed(sum, a, curve_b, (er >> 31) != 0);
    }
    return sum;
}

// sum of the stretches of bucket b: slots b + first/chunk .. b + (first + count - 1)/chunk; stretch q holds the entries
// [max(first, q chunk), min(first + count, (q + 1) chunk)) of the window's run
template <class C, class Source, class Points>
ECGPU_HD Proj<C> msm_stretch_of(uint32_t b, uint32_t q, uint32_t first, uint32_t count, uint32_t chunk, const Fe<C::NL>& curve_b,
                                const Source& partial, const uint32_t* __restrict__ run, const Points& points) {
    const uint32_t c0 = q * chunk, c1 = c0 + chunk;
    const uint32_t lo = first > c0 ? first : c0, hi = first + count < c1 ? first + count : c1;
    return msm_stretch_sum<C>(partial.get((size_t)b + q), run, lo, hi, curve_b, points);
}
template <class C, class Source, class Points>
ECGPU_HD Proj<C> msm_bucket_finish(uint32_t b, uint32_t first, uint32_t count, uint32_t chunk, const Fe<C::NL>& curve_b,
                                   const Source& partial, const uint32_t* __restrict__ run, const Points& points) {
    using G = Group<C>;
    if (count == 0) return G::identity();
    const uint32_t q0 = first / chunk, q1 = (first + count - 1) / chunk;
    Proj<C> acc = msm_stretch_of<C>(b, q0, first, count, chunk, curve_b, partial, run, points);
#pragma unroll 1
    for (uint32_t q = q0 + 1; q <= q1; q++)
        acc = G::add(acc, msm_stretch_of<C>(b, q, first, count, chunk, curve_b, partial, run, points), curve_b);
    return acc;
}

}  // namespace ecgpu
