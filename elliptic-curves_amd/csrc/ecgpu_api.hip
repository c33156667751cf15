// ecgpu_api.hip — the C ABI of include/ecgpu.h on top of the gfx950 kernels.
// No torch, no CPU compute path: every entry point either runs HIP kernels or returns an error.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/ecgpu.h"
#include "ecgpu_launch.h"
#include "ecgpu_knobs.h"
#include "ecgpu_recode.h"

using namespace ecgpu;

#ifndef ECGPU_FIXED_SOA_DEFAULT
#define ECGPU_FIXED_SOA_DEFAULT true
#endif

namespace {

constexpr int BLOCK = 256;
enum : int { ST_BAD_SCALAR = 1, ST_BAD_POINT = 2 };

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    size_t dirty = 0;         // bytes from the start that calls have asked for since the buffer was last wiped (wipe_scratch zeroes these, not cap)
};

// A context's view of a basepoint comb table.  The table itself is owned by the per-device registry below and shared by
// every context of the process on that device (the analogue of the reference's process-wide `LazyLock<BasepointTable>`,
// k256/src/arithmetic/tables.rs:18): the 21.5 GB k256 table is built and held once per GPU however many contexts exist.
struct Table {
    uint32_t* d = nullptr;
    int w = 0, nwin = 0;      // the width actually in use (may be narrower than asked for after an out-of-memory fallback)
    int asked = 0;            // the width this view was made for (ctx->want_w at that time)
};

struct SharedTable {
    uint32_t* d = nullptr;
    int nwin = 0, refs = 0;
    size_t bytes = 0;         // device memory of the table
    double build_ms = 0;      // host wall time of its construction (allocation, kernels, the wait for them)
};
struct TableRegistry {
    std::mutex mu;                                               // held while a table is built: a second context of the device waits
    std::map<std::tuple<int, int, int>, SharedTable> tabs;       // (device, curve id, comb width)
    // widths whose allocation was refused -> calls left before the width is tried again: later contexts go straight to the
    // width that worked, but a refusal is not for ever (the memory may have been another process's or torch's cache);
    // forgotten at once when a table of the device is freed or enough memory shows as free
    std::map<std::tuple<int, int, int>, int> nofit;
    uint64_t seen[12] = {};                                      // generator multiplications asked of this device so far, per curve
};
constexpr int NOFIT_RETRY_CALLS = 64;
// adaptive policy: generator multiplications (log2) after which a device moves from the 16-bit table to the 22-bit one and from
// there to the context's widest (table_tier below)
constexpr int TABLE_TIER1_LOG2 = 26, TABLE_TIER2_LOG2 = 29;
// test-only fault injection (tests/test_gpu_multidevice.py through the exported ecgpu_testhook_table_max_mb; no environment
// variable: the production path cannot be steered from outside the process): comb tables above this many MiB are refused
std::atomic<size_t> g_test_table_max_mb{0};
TableRegistry& table_registry(int device) {                      // one per device: the GPUs of a group build in parallel
    static TableRegistry* r = new TableRegistry[64];             // never destroyed: contexts may outlive static destructors
    return r[device & 63];
}

}  // namespace

struct ecgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t up_stream = nullptr, down_stream = nullptr;   // host <-> device legs of the pipelined host-pointer calls
    std::string err;
    int* d_status = nullptr;
    int* h_status = nullptr;
    Table table[12];
    uint32_t* ct_lut[12] = {};   // uniform-schedule generator LUTs (shared per device like the comb tables; registry width key -1)
    // fixed-base comb width: every addition removed is worth 8 % and HBM keeps up with the gathers, so the tables are
    // sized for 288 GB, not for a cache.  k256: W = 26, 10 windows = 9 additions per scalar, 21.5 GB, built in 65 ms;
    // p256 and sm2: W = 24, 11 windows, 5.9 GB; p384: W = 20, 1.0 GB.  ecgpu_set_base_window trades memory for speed.
    int want_w[12] = {26, 24, 20, 24, 24, 24, 20, 24, 20, 24, 20, 24};
    // Footprint policy (ecgpu_set_table_policy / ecgpu_set_table_budget / ecgpu_set_base_window): want_w is the WIDEST table a
    // context will use; unless the width is pinned or the policy is eager, the table grows with the number of generator
    // multiplications the device has been asked for (table_tier) — the reference builds its 30-60 KB tables lazily and prices
    // a table half the size at 3 % (k256/src/arithmetic/mul.rs:191-192, primeorder/src/tables/basepoint.rs:29-76)
    bool w_pinned[12] = {};
    int table_policy = 0;        // ECGPU_TABLE_ADAPTIVE
    size_t table_budget = 0;     // bytes one comb table may take; 0 = no limit
    int msm_c = 0;   // 0 = choose from n
    DevBuf proj, prefix, vtab, bases, in0, in1, in2, in3, out0, out1, msm_ws;
    DevBuf ec_u1, ec_u2, ec_q, ec_valid, ec_xy, ec_inf, ec_r, ec_e, ec_s, ec_id;   // signature verification scratch
    DevBuf ec_winv;                    // the batch's s^-1 / r^-1 modulo the group order (k_scalar_batch_inv; its prefix products use `prefix`)
    DevBuf ct_flags;             // one verdict byte per element of a uniform-schedule batch
    DevBuf cx_xy, cx_inf;        // x || y + flag records decoded from compressed input (ecgpu_msm_compressed, ecgpu_batch_mul_compressed)
    bool keep_status = false;    // the status word already holds the verdicts of a first stage of the call: do not clear it
    hipEvent_t ev[9] = {};       // 0..2 call spans, 3..4 the MSM's sort / accumulate marks, 5 spare, 6..8 MsmPlan::detail
    std::map<std::string, double> timing;
    std::vector<std::pair<std::string, std::pair<int, int>>> spans;   // event pairs of the last call not yet turned into `timing`
    // asynchronous mode (ecgpu_set_async): device-pointer calls return once their work is queued; the status word
    // accumulates on the device until ecgpu_synchronize (or a host-pointer call) collects it into `deferred`
    bool async = false, pending = false;
    bool timing_on = true;       // ecgpu_set_timing: per-call HIP events (ecgpu_last_timing)
    int deferred = 0;
    // MSM lanes (ecgpu_set_msm_lanes): in asynchronous mode consecutive MSMs alternate between two internal streams, each with
    // a workspace of its own, so that the sort and the reduction tail of one MSM (bandwidth- and latency-bound) run beside the
    // accumulation of the other (issue-bound)
    struct MsmLane {
        hipStream_t s = nullptr;
        DevBuf ws, proj, prefix;
        hipEvent_t ev_in = nullptr, ev_a = nullptr, ev_b = nullptr, ev_done = nullptr;
        const void* parts_out = nullptr;   // the parts record an ecgpu_msm_parts_dev on this lane wrote last (ecgpu_msm_parts_join_dev)
    };
    bool lanes_pending = false;  // an MSM was queued on a lane since the last other call: that call first waits for the lanes (ev_done)
    int msm_lanes = 1;
    unsigned msm_seq = 0;
    MsmLane lane[4];
    int lane_last = -1;          // the lane of the last MSM queued on one (ecgpu_last_timing "accumulate" reads its events)
};

namespace {

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return e_ == hipErrorOutOfMemory ? ECGPU_ERR_OOM : ECGPU_ERR_HIP;                   \
        }                                                                                       \
    } while (0)

int ensure(ecgpu_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes > b.dirty) b.dirty = bytes;
    if (bytes <= b.cap) return ECGPU_OK;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return ECGPU_OK;
}

// the same for a buffer that is used on another stream of the context (an MSM lane)
int ensure_on(ecgpu_ctx* ctx, hipStream_t stream, DevBuf& b, size_t bytes) {
    if (bytes > b.dirty) b.dirty = bytes;
    if (bytes <= b.cap) return ECGPU_OK;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return ECGPU_OK;
}

template <class F>
int dispatch(int curve, F&& f) {
    switch (curve) {
    case ECGPU_K256: return f(K256Params{});
    case ECGPU_P256: return f(P256Params{});
    case ECGPU_P384: return f(P384Params{});
    case ECGPU_SM2: return f(Sm2Params{});
    case ECGPU_P224: return f(P224Params{});
    case ECGPU_P192: return f(P192Params{});
    case ECGPU_P521: return f(P521Params{});
    case ECGPU_BP256: return f(Bp256Params{});
    case ECGPU_BP384: return f(Bp384Params{});
    case ECGPU_BP256T1: return f(Bp256t1Params{});
    case ECGPU_BP384T1: return f(Bp384t1Params{});
    case ECGPU_BIGN256: return f(Bign256Params{});
    default: return ECGPU_ERR_CURVE;
    }
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// MSMs in flight on the lanes: whatever the context does next on its own stream is ordered after them
int join_lanes(ecgpu_ctx* ctx) {
    if (!ctx->lanes_pending) return ECGPU_OK;
    for (auto& l : ctx->lane)
        if (l.s && l.ev_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, l.ev_done, 0));
    ctx->lanes_pending = false;
    return ECGPU_OK;
}

int reset_status(ecgpu_ctx* ctx) {
    int rc = join_lanes(ctx);
    if (rc != ECGPU_OK) return rc;
    if (ctx->async || ctx->keep_status) return ECGPU_OK;          // flags accumulate until ecgpu_synchronize / the call's end
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    return ECGPU_OK;
}

int status_error(ecgpu_ctx* ctx, int st) {
    if (st & ST_BAD_SCALAR) { ctx->err = "scalar not in [0, n)"; return ECGPU_ERR_SCALAR_RANGE; }
    if (st & ST_BAD_POINT) { ctx->err = "point coordinate >= p or not on curve"; return ECGPU_ERR_POINT; }
    return ECGPU_OK;
}

// reads the status word back (synchronises the stream) and maps it to an error code; asynchronous mode: returns at once,
// the word is read by drain()
int finish(ecgpu_ctx* ctx) {
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->async) {
        ctx->pending = true;
        return ECGPU_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return status_error(ctx, *ctx->h_status);
}

// asynchronous mode: wait for the queued work, move its status flags into ctx->deferred, clear the device word
int drain(ecgpu_ctx* ctx) {
    for (auto& l : ctx->lane)                  // queued MSMs on the lanes: their status flags land in the same word
        if (l.s) HIP_TRY(ctx, hipStreamSynchronize(l.s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->deferred |= *ctx->h_status;
    ctx->pending = false;
    return ECGPU_OK;
}

// A host-pointer call on an asynchronous context runs synchronously: it first waits for the queued work (whose errors stay
// deferred for ecgpu_synchronize), and reports its own errors itself.
struct SyncScope {
    ecgpu_ctx* ctx;
    bool was;
    int rc = ECGPU_OK;      // a failed drain (sticky device fault of the queued work, failed copy): the entry point returns it
    explicit SyncScope(ecgpu_ctx* c) : ctx(c), was(c->async) {
        if (was) {
            rc = drain(ctx);
            ctx->async = false;
        }
    }
    ~SyncScope() {
        if (was) {
            (void)hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream);
            ctx->async = true;
        }
    }
};

// The uniform-schedule entry points are meant for secret scalars, and the reference keeps such values in zeroize-on-drop
// types (`NonZeroScalar`, `SharedSecret`).  Whatever path such a call leaves by, the context's scratch that held values derived
// from its secrets — k P in projective form, the running products of the batch inversion, the affine products of an ECDH —
// is zeroed behind its last kernel (stream-ordered), and `staging` also clears the copies a host-pointer call made of the
// caller's scalars and results.  The caller's own buffers are the caller's to wipe; ecgpu_wipe does the same on request.
enum : int { WIPE_SCRATCH = 1, WIPE_EC = 2, WIPE_STAGING = 4 };
void wipe_scratch(ecgpu_ctx* ctx, int what) {
    auto clear = [&](std::initializer_list<DevBuf*> bufs) {
        // what has been asked of a buffer since its last wipe, not its capacity: after one 2^20-term batch the scratch holds
        // ~170 MB, and zeroing all of it behind every later 1,024-scalar `_ct` call (or every chunk of a pipelined one) cost tens
        // of microseconds per call
        for (DevBuf* b : bufs) {
            if (b->p && b->dirty) (void)hipMemsetAsync(b->p, 0, b->dirty < b->cap ? b->dirty : b->cap, ctx->stream);
            b->dirty = 0;
        }
    };
    if (what & WIPE_SCRATCH) clear({&ctx->proj, &ctx->prefix});
    if (what & WIPE_EC) clear({&ctx->ec_xy, &ctx->ec_inf});
    if (what & WIPE_STAGING) clear({&ctx->in0, &ctx->in3, &ctx->out0, &ctx->out1});
}
struct CtWipe {
    ecgpu_ctx* ctx;
    int what;
    CtWipe(ecgpu_ctx* c, int what_) : ctx(c), what(what_) {}
    ~CtWipe() {
        if (what) wipe_scratch(ctx, what);
    }
};

void record(ecgpu_ctx* ctx, int i) {
    if (ctx->timing_on) (void)hipEventRecord(ctx->ev[i], ctx->stream);
}

void resolve_timing(ecgpu_ctx* ctx) {
    if (ctx->spans.empty()) return;
    ctx->timing.clear();
    for (auto& s : ctx->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev[s.second.first], ctx->ev[s.second.second]) == hipSuccess)
            ctx->timing[s.first] = ms;
        else
            (void)hipGetLastError();         // a mark the call never recorded (an MSM of no terms): no span, and no error left behind
    }
    ctx->spans.clear();
}

// the spans of the call just made; turned into milliseconds now, or (asynchronous mode: the events have not happened
// yet) when ecgpu_last_timing asks
void collect_timing(ecgpu_ctx* ctx, std::initializer_list<std::pair<const char*, std::pair<int, int>>> spans) {
    ctx->spans.clear();
    ctx->lane_last = -1;
    if (!ctx->timing_on) {                     // no events were recorded for this call: ecgpu_last_timing has nothing to report
        ctx->timing.clear();
        return;
    }
    for (auto& s : spans) ctx->spans.emplace_back(s.first, s.second);
    if (!ctx->async) resolve_timing(ctx);
}

// ---- basepoint table ---------------------------------------------------------------------------------

// drops this context's reference to its table of curve `id`; the last reference frees the device memory
void release_table(ecgpu_ctx* ctx, int id) {
    Table& t = ctx->table[id];
    if (!t.d) return;
    TableRegistry& reg = table_registry(ctx->device);
    std::lock_guard<std::mutex> lock(reg.mu);
    auto it = reg.tabs.find(std::make_tuple(ctx->device, id, t.w));
    if (it != reg.tabs.end() && it->second.d == t.d && --it->second.refs == 0) {
        (void)hipFree(it->second.d);
        reg.tabs.erase(it);
        reg.nofit.clear();                    // memory came back: widths refused earlier may fit now
    }
    t = Table();
}

// builds the comb table of width w into freshly allocated device memory; ECGPU_ERR_OOM when the table (or its build
// scratch) does not fit
template <class C>
int build_table(ecgpu_ctx* ctx, int w, SharedTable* out) {
    constexpr int N = C::N, NS = Field<C>::NS;
    const int bits = 32 * N;
    const int nwin = signed_window_count(bits - 1, w);       // scalars are folded to bits - 1 bits (fold_scalar)
    const size_t half = (size_t)1 << (w - 1);
    const size_t entries = half * nwin;
    int rc;
    // windows are built in slabs so that the projective scratch (192 B per entry, 3x the table) stays below ~2 GB
    size_t slab = ((size_t)2 << 30) / (half * (4 * NS) * 4);
    if (slab < 1) slab = 1;
    if (slab > (size_t)nwin) slab = nwin;
    if ((rc = ensure(ctx, ctx->bases, (size_t)nwin * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->proj, slab * half * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->prefix, slab * half * NS * 4)) != ECGPU_OK) return rc;
    uint32_t* d = nullptr;
    const auto t_build0 = std::chrono::steady_clock::now();
    if (const size_t cap_mb = g_test_table_max_mb.load()) {     // fault injection for the fallback path (ecgpu_testhook_table_max_mb)
        if (entries * 2 * N * 4 > cap_mb << 20) {
            ctx->err = "basepoint table: allocation refused by the test hook";
            return ECGPU_ERR_OOM;
        }
    }
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), entries * 2 * N * 4));
    launch_window_bases<C>(ctx->stream, (uint32_t*)ctx->bases.p, w, nwin);
    for (size_t j0 = 0; j0 < (size_t)nwin; j0 += slab) {
        const size_t ws = j0 + slab <= (size_t)nwin ? slab : (size_t)nwin - j0;
        launch_table_entries<C>(ctx->stream, (const uint32_t*)ctx->bases.p + j0 * (3 * NS), (uint32_t*)ctx->proj.p, w, (int)ws);
        launch_normalize<C>(ctx->stream, true, (const uint32_t*)ctx->proj.p, (uint32_t*)ctx->prefix.p, ws * half, nullptr,
                            nullptr, d + j0 * half * (2 * N));
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // visible to every stream of the device from here on
    if (e != hipSuccess) {
        (void)hipFree(d);
        ctx->err = std::string("basepoint table build: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? ECGPU_ERR_OOM : ECGPU_ERR_HIP;
    }
    out->d = d;
    out->nwin = nwin;
    out->bytes = entries * 2 * N * 4;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    return ECGPU_OK;
}

// the build scratch is larger than most batches need: give it back
int drop_build_scratch(ecgpu_ctx* ctx) {
    for (DevBuf* b : {&ctx->proj, &ctx->prefix}) {
        if (b->cap > ((size_t)64 << 20)) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(b->p));
            b->p = nullptr;
            b->cap = 0;
        }
    }
    return ECGPU_OK;
}

// bytes of the comb table of width w for curve C
template <class C>
size_t comb_table_bytes(int w) {
    return ((size_t)1 << (w - 1)) * (size_t)signed_window_count(32 * C::N - 1, w) * 2 * C::N * 4;
}

// The width the adaptive policy gives a device that has been asked for `seen` generator multiplications of a curve so far (the
// call being served included), capped by `wmax`.  Every step up is taken where the time the narrower table has cost so far equals
// the time the next table takes to build (a rent-or-buy rule: never more than twice the time of the best fixed choice, whatever
// the caller goes on to do) — measured on MI355X for k256, profiles/r05/table_tiers.txt: 16 bits = 34 MB built in 1 ms, 9
// windows more per scalar than at 26; 22 bits = 1.6 GB in 5 ms; 26 bits = 21.5 GB in 65 ms.
inline int table_tier(uint64_t seen, int wmax) {
    int w = seen < ((uint64_t)1 << TABLE_TIER1_LOG2) ? 16 : seen < ((uint64_t)1 << TABLE_TIER2_LOG2) ? 22 : wmax;
    return w < wmax ? w : wmax;
}

// Makes ctx->table[C::ID] point at the device's shared comb table for a call with n scalars: the width is the context's pinned
// width (ecgpu_set_base_window), the widest allowed one (eager policy) or the tier the device's history asks for (adaptive, the
// default), never above the budget (ecgpu_set_table_budget) — and a wider table another context of the device has already built
// is taken as it is.  The table is built if no context of this process has yet.  When it does not fit — the k256 maximum is
// 21.5 GB — the width is lowered two bits at a time (a quarter of the memory, one or two more additions per scalar; results
// do not depend on it) down to 16 bits (36 MB) before ECGPU_ERR_OOM is returned.
template <class C>
int ensure_table(ecgpu_ctx* ctx, size_t n = 0) {
    Table& t = ctx->table[C::ID];
    TableRegistry& reg = table_registry(ctx->device);
    std::unique_lock<std::mutex> lock(reg.mu);
    const int wmax = ctx->want_w[C::ID];
    uint64_t& seen = reg.seen[C::ID];
    seen = seen + n < seen ? ~(uint64_t)0 : seen + n;
    // the width this call should get, from the registry as it is NOW (called again whenever the lock was dropped)
    const auto choose = [&]() -> int {
        int want = wmax;
        if (ctx->w_pinned[C::ID]) return want;
        if (ctx->table_policy == ECGPU_TABLE_ADAPTIVE) want = table_tier(seen, wmax);
        if (ctx->table_budget)
            while (want > 4 && comb_table_bytes<C>(want) > ctx->table_budget) want--;
        for (int w = wmax; w > want; w--) {                      // somebody has paid for a wider one already
            auto it = reg.tabs.find(std::make_tuple(ctx->device, (int)C::ID, w));
            if (it != reg.tabs.end() && it->second.d && (!ctx->table_budget || it->second.bytes <= ctx->table_budget)) return w;
        }
        return want;
    };
    // a refusal of `w` on this device is on record: has it expired (memory back, or NOFIT_RETRY_CALLS calls old)?  Erases it if so.
    const auto refusal_over = [&](int w) -> bool {
        auto nf = reg.nofit.find(std::make_tuple(ctx->device, (int)C::ID, w));
        if (nf == reg.nofit.end()) return true;
        size_t free_b = 0, total_b = 0;
        const bool roomy = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > comb_table_bytes<C>(w) + ((size_t)3 << 30) &&
                           !g_test_table_max_mb.load();
        if (!roomy && --nf->second > 0) return false;
        reg.nofit.erase(nf);
        return true;
    };
    int want = choose();
    if (t.d && t.asked == want) {
        if (t.w >= want) return ECGPU_OK;
        // The context sits on a NARROWER table than it asked for (the wide one did not fit when it was built).  That refusal is not
        // for ever (include/ecgpu.h): when it has expired the wide table is tried again — before the narrow one is let go, so that
        // a second refusal costs one failed allocation and nothing else.
        if (!refusal_over(want)) return ECGPU_OK;
        const auto key = std::make_tuple(ctx->device, (int)C::ID, want);
        SharedTable& st = reg.tabs[key];
        if (!st.d) {
            const int rc = build_table<C>(ctx, want, &st);
            if (rc != ECGPU_OK) {
                reg.tabs.erase(key);
                (void)hipGetLastError();
                (void)drop_build_scratch(ctx);
                if (rc != ECGPU_ERR_OOM) return rc;
                reg.nofit[key] = NOFIT_RETRY_CALLS;              // still no room: stay on the narrow table for another stretch of calls
                return ECGPU_OK;
            }
        }
        st.refs++;                                               // ours from here on: it cannot go away while the lock is dropped
        const Table fresh{st.d, want, st.nwin, want};
        lock.unlock();
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));         // queued kernels may still read the narrow table
        release_table(ctx, C::ID);
        t = fresh;
        return drop_build_scratch(ctx);
    }
    if (t.d) {
        lock.unlock();
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // queued kernels may still read the old table
        release_table(ctx, C::ID);
        lock.lock();
        want = choose();      // the registry may have changed meanwhile: a wider table this call meant to share can be gone with its last owner
    }
    int rc = ECGPU_ERR_OOM;
    for (int w = want;; w -= 2) {
        const auto key = std::make_tuple(ctx->device, (int)C::ID, w);
        if (w - 2 >= 16 && !refusal_over(w)) continue;           // an earlier context of the device was refused this width
        reg.nofit.erase(key);
        SharedTable& st = reg.tabs[key];
        if (!st.d) {
            rc = build_table<C>(ctx, w, &st);
            if (rc != ECGPU_OK) {
                reg.tabs.erase(key);
                (void)hipGetLastError();                         // an out-of-memory error is not sticky
                int rc2 = drop_build_scratch(ctx);
                if (rc == ECGPU_ERR_OOM) reg.nofit[key] = NOFIT_RETRY_CALLS;
                if (rc == ECGPU_ERR_OOM && rc2 == ECGPU_OK && w - 2 >= 16) continue;
                if (rc == ECGPU_ERR_OOM) ctx->err = "basepoint comb table does not fit in device memory (even at 16-bit windows)";
                return rc;
            }
        }
        st.refs++;
        t.d = st.d;
        t.w = w;
        t.nwin = st.nwin;
        t.asked = want;
        break;
    }
    return drop_build_scratch(ctx);
}

// ---- generator LUTs of the uniform-schedule fixed-base kernel (ecgpu_ctmul.h) ----------------------------------------
// [CT_BASE_LUTS][CT_BASE_ENTRIES][2] packed elements: lut i = {e * 2^(W i) * G, e = 1..2^(W-1)}, W = CT_BASE_W — `BasepointTable::new`
// (primeorder/src/tables/basepoint.rs:41-76) with affine entries.  43 LUTs of 32 entries = 88 KB for k256: built with the
// comb-table kernels (bases 2^(6 i) G, 32 multiples each, one normalisation), shared per device under the registry key width -1.
void release_ct_lut(ecgpu_ctx* ctx, int id) {
    if (!ctx->ct_lut[id]) return;
    TableRegistry& reg = table_registry(ctx->device);
    std::lock_guard<std::mutex> lock(reg.mu);
    auto it = reg.tabs.find(std::make_tuple(ctx->device, id, -1));
    if (it != reg.tabs.end() && it->second.d == ctx->ct_lut[id] && --it->second.refs == 0) {
        (void)hipFree(it->second.d);
        reg.tabs.erase(it);
    }
    ctx->ct_lut[id] = nullptr;
}

template <class C>
int ensure_ct_lut(ecgpu_ctx* ctx) {
    if (ctx->ct_lut[C::ID]) return ECGPU_OK;
    constexpr int N = C::N, NS = Field<C>::NS;
    const int nlut = ct_base_luts<C>();
    const size_t entries = (size_t)nlut * CT_BASE_ENTRIES;
    TableRegistry& reg = table_registry(ctx->device);
    std::lock_guard<std::mutex> lock(reg.mu);
    SharedTable& st = reg.tabs[std::make_tuple(ctx->device, (int)C::ID, -1)];
    if (!st.d) {
        int rc;
        auto fail = [&](int code) {
            reg.tabs.erase(std::make_tuple(ctx->device, (int)C::ID, -1));
            return code;
        };
        if ((rc = ensure(ctx, ctx->bases, (size_t)nlut * 3 * NS * 4)) != ECGPU_OK) return fail(rc);
        if ((rc = ensure(ctx, ctx->proj, entries * 3 * NS * 4)) != ECGPU_OK) return fail(rc);
        if ((rc = ensure(ctx, ctx->prefix, entries * NS * 4)) != ECGPU_OK) return fail(rc);
        uint32_t* d = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d), entries * 2 * N * 4) != hipSuccess) {
            (void)hipGetLastError();
            ctx->err = "generator LUTs: hipMalloc failed";
            return fail(ECGPU_ERR_OOM);
        }
        launch_window_bases<C>(ctx->stream, (uint32_t*)ctx->bases.p, CT_BASE_W, nlut);                  // 2^(W i) G
        launch_table_entries<C>(ctx->stream, (const uint32_t*)ctx->bases.p, (uint32_t*)ctx->proj.p, CT_BASE_W, nlut);   // e = 1..32
        launch_normalize<C>(ctx->stream, true, (const uint32_t*)ctx->proj.p, (uint32_t*)ctx->prefix.p, entries, nullptr, nullptr, d);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(d);
            ctx->err = std::string("generator LUT build: ") + hipGetErrorString(e);
            return fail(ECGPU_ERR_HIP);
        }
        st.d = d;
        st.nwin = nlut;
    }
    st.refs++;
    ctx->ct_lut[C::ID] = st.d;
    return ECGPU_OK;
}

// launches the normalisation of n projective points in ctx->proj to wire-format output
template <class C>
int normalize_out(ecgpu_ctx* ctx, size_t n, void* d_out_xy, void* d_out_inf, bool soa = false) {
    int rc;
    if ((rc = ensure(ctx, ctx->prefix, n * Field<C>::NS * 4)) != ECGPU_OK) return rc;
    launch_normalize<C>(ctx->stream, false, (const uint32_t*)ctx->proj.p, (uint32_t*)ctx->prefix.p, n, (uint8_t*)d_out_xy,
                        (uint8_t*)d_out_inf, nullptr, soa);
    return ECGPU_OK;
}

// The hand-over between k_fixed_base and k_normalize quad-major (store_proj_soa, ecgpu_kernels.h: a wave's load or store is 1,024
// contiguous bytes instead of 64 pieces 144 bytes apart)?  A/B: ECGPU_FIXED_SOA = 0 / 1 in the tool build (profiles/r06/).
inline bool fixed_soa() {
    if (const char* e = knob("ECGPU_FIXED_SOA")) return e[0] != '0';
    return ECGPU_FIXED_SOA_DEFAULT;
}

// ---- device-pointer implementations --------------------------------------------------------------------

template <class C>
int mul_base_dev(ecgpu_ctx* ctx, const void* d_scalars, size_t n, void* d_out_xy, void* d_out_inf, bool compressed = false) {
    constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
    int rc;
    if ((rc = ensure_table<C>(ctx, n)) != ECGPU_OK) return rc;
    if (n == 0) return ECGPU_OK;
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    const Table& t = ctx->table[C::ID];
    record(ctx, 0);
    const bool soa = !compressed && fixed_soa();
    launch_fixed_base<C>(ctx->stream, (const uint8_t*)d_scalars, n, (const uint32_t*)t.d, t.w, t.nwin, (uint32_t*)ctx->proj.p,
                         ctx->d_status, soa);
    record(ctx, 1);
    if (compressed) {
        if ((rc = ensure(ctx, ctx->prefix, n * NS * 4)) != ECGPU_OK) return rc;
        launch_normalize_compressed<C>(ctx->stream, (const uint32_t*)ctx->proj.p, (uint32_t*)ctx->prefix.p, n, (uint8_t*)d_out_xy,
                                       (uint8_t*)d_out_inf);
    } else if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf, soa)) != ECGPU_OK) {
        return rc;
    }
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

template <class C>
int mul_var_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_points_xy, const void* d_points_inf, size_t n,
                void* d_out_xy, void* d_out_inf) {
    constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
    if (n == 0) return ECGPU_OK;
    int rc;
    size_t tstride = var_base_slots<C>(n);
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_var_base<C>(ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_points_xy, (const uint8_t*)d_points_inf, n,
                       (uint32_t*)ctx->vtab.p, tstride, (uint32_t*)ctx->proj.p, ctx->d_status);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

// ---- uniform-schedule variants (ecgpu_ct.h): the reference's constant-time drivers as they are ---------------------------
template <class C>
int mul_base_ct_dev(ecgpu_ctx* ctx, const void* d_scalars, size_t n, void* d_out_xy, void* d_out_inf) {
    constexpr int NS = Field<C>::NS;
    int rc;
    if ((rc = ensure_ct_lut<C>(ctx)) != ECGPU_OK) return rc;
    if (n == 0) return ECGPU_OK;
    CtWipe wipe(ctx, WIPE_SCRATCH);
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ct_flags, n + 16)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_fixed_base_ct<C>(ctx->stream, (const uint8_t*)d_scalars, n, (const uint32_t*)ctx->ct_lut[C::ID], (uint32_t*)ctx->proj.p,
                            (uint8_t*)ctx->ct_flags.p, ctx->d_status);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

template <class C>
int mul_var_ct_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_points_xy, const void* d_points_inf, size_t n,
                   void* d_out_xy, void* d_out_inf) {
    constexpr int NS = Field<C>::NS;
    if (n == 0) return ECGPU_OK;
    int rc;
    CtWipe wipe(ctx, WIPE_SCRATCH);
    size_t tstride = var_base_slots<C>(n);
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ct_flags, n + 16)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_var_base_ct<C>(ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_points_xy, (const uint8_t*)d_points_inf, n,
                          (uint32_t*)ctx->vtab.p, tstride, (uint32_t*)ctx->proj.p, (uint8_t*)ctx->ct_flags.p, ctx->d_status);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

template <class C>
int normalize_dev(ecgpu_ctx* ctx, const void* d_xyz, size_t n, void* d_out_xy, void* d_out_inf) {
    constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_load_proj<C>(ctx->stream, (const uint8_t*)d_xyz, n, (uint32_t*)ctx->proj.p, ctx->d_status);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

template <class C>
int point_sum_dev(ecgpu_ctx* ctx, const void* d_xy, const void* d_inf, size_t n, void* d_out_xy, void* d_out_inf) {
    constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
    int rc;
    if ((rc = ensure(ctx, ctx->proj, 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_point_sum<C>(ctx->stream, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n, (uint32_t*)ctx->proj.p, ctx->d_status);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, 1, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

// The lane the next MSM (or local half of a sharded MSM) of an asynchronous context with ecgpu_set_msm_lanes > 1 goes to: lanes take
// turns; a lane's stream, events and workspace exist from its first use on.
int next_lane(ecgpu_ctx* ctx, size_t workspace_bytes, ecgpu_ctx::MsmLane** out) {
    ctx->lane_last = (int)(ctx->msm_seq % (unsigned)ctx->msm_lanes);
    ecgpu_ctx::MsmLane& l = ctx->lane[ctx->msm_seq++ % (unsigned)ctx->msm_lanes];
    ctx->spans.clear();
    ctx->timing.clear();
    if (!l.s) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&l.ev_in, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreate(&l.ev_a));
        HIP_TRY(ctx, hipEventCreate(&l.ev_b));
        HIP_TRY(ctx, hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming));
    }
    int rc = ensure_on(ctx, l.s, l.ws, workspace_bytes);
    if (rc != ECGPU_OK) return rc;
    *out = &l;
    return ECGPU_OK;
}

// largest term count for which the per-term multiplication + tree sum replaces the bucket method (0: never);
// ECGPU_MSM_SMALL_LOG2 overrides the measured default (tuning knob, -1 disables)
template <class C>
size_t msm_small_max() {
    // measured (round 1, tools/gpu_msm_sweep.py is today's form of the sweep): k256 0.75-0.91 ms against 1.21-1.36 ms up to 2^16 terms (1.51 against 1.39 at 2^17),
    // p256 1.23-1.44 against 1.39-1.59 ms, p384 3.3-3.4 against 3.7-4.1 ms up to 2^10 and level beyond
    int lg = C::N > 8 ? 10 : 16;
    if (const char* e = knob("ECGPU_MSM_SMALL_LOG2")) lg = atoi(e);
    return lg <= 0 ? 0 : (size_t)1 << (lg > 24 ? 24 : lg);
}

template <class C>
int msm_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_xy, const void* d_inf, size_t n, void* d_out_xy,
            void* d_out_inf) {
    constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
    int rc;
    if (n > msm_max_terms<C>()) {           // sorted entries are sub-term index | sign << 31
        ctx->err = "MSM of 2^31 (k256: 2^30) or more terms: split it and add the partial sums (ecgpu_point_sum)";
        return ECGPU_ERR_ARG;
    }
    if (n >= 1 && n <= msm_small_max<C>() && ctx->msm_c == 0) {
        // Small MSM: the bucket method has a floor of ~1.2 ms of serial work that does not depend on n (running sums, the
        // 240-doubling combine chain).  Below ~2^17 terms one variable-base multiplication per term (all lanes in
        // parallel, ~0.5 ms of latency) and a tree sum of the products are faster.
        size_t tstride = var_base_slots<C>(n);
        if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->prefix, ((n + BLOCK - 1) / BLOCK + 1) * 3 * NS * 4)) != ECGPU_OK) return rc;
        if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
        record(ctx, 0);
        launch_var_base<C>(ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n,
                           (uint32_t*)ctx->vtab.p, tstride, (uint32_t*)ctx->proj.p, ctx->d_status);
        record(ctx, 3);
        launch_proj_sum<C>(ctx->stream, (uint32_t*)ctx->proj.p, n, (uint32_t*)ctx->prefix.p);
        record(ctx, 4);
        record(ctx, 1);
        if ((rc = normalize_out<C>(ctx, 1, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
        record(ctx, 2);
        rc = finish(ctx);
        collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}, {"sort", {0, 3}},
                             {"accumulate", {3, 4}}, {"reduce", {4, 1}}});
        return rc;
    }
    MsmPlan plan = msm_plan<C>(n, ctx->msm_c, msm_use_glv<C>(n));
    if (ctx->async && ctx->msm_lanes > 1) {
        // one of the lanes: everything of this MSM on the lane's stream and in the lane's buffers, ordered after what the
        // context's stream holds now (the inputs); its output is ordered by ecgpu_synchronize only
        ecgpu_ctx::MsmLane* lp = nullptr;
        if ((rc = next_lane(ctx, plan.workspace_bytes, &lp)) != ECGPU_OK) return rc;
        ecgpu_ctx::MsmLane& l = *lp;
        l.parts_out = nullptr;
        if ((rc = ensure_on(ctx, l.s, l.proj, 3 * NS * 4)) != ECGPU_OK) return rc;
        if ((rc = ensure_on(ctx, l.s, l.prefix, NS * 4)) != ECGPU_OK) return rc;
        HIP_TRY(ctx, hipEventRecord(l.ev_in, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(l.s, l.ev_in, 0));
        launch_msm<C>(plan, l.s, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n, l.ws.p, (uint32_t*)l.proj.p,
                      ctx->d_status, l.ev_a, l.ev_b, (uint8_t*)d_out_xy, (uint8_t*)d_out_inf);   // (the last kernel writes the wire record)
        // Any OTHER entry point called later on this context waits for this event first (reset_status): it may read the
        // output or reuse the inputs.  Further MSMs do not — they go to the next lane — and neither does work the caller
        // queues on the stream itself: for that, inputs and outputs belong to the lane until ecgpu_synchronize.
        HIP_TRY(ctx, hipEventRecord(l.ev_done, l.s));
        ctx->lanes_pending = true;
        return finish(ctx);
    }
    if ((rc = ensure(ctx, ctx->proj, 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->msm_ws, plan.workspace_bytes)) != ECGPU_OK) return rc;
    const bool detail = ctx->timing_on && ctx->ev[6];
    if (detail)
        for (int i = 0; i < 3; i++) plan.detail[i] = ctx->ev[6 + i];
    record(ctx, 0);
    launch_msm<C>(plan, ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n,
                  ctx->msm_ws.p, (uint32_t*)ctx->proj.p, ctx->d_status, ctx->timing_on ? ctx->ev[3] : nullptr, ctx->timing_on ? ctx->ev[4] : nullptr, (uint8_t*)d_out_xy,
                  (uint8_t*)d_out_inf);
    record(ctx, 1);     // (the conversion to affine happens inside the last kernel of the chain: "normalize" is an empty span)
    record(ctx, 2);
    rc = finish(ctx);
    // "accumulate" is the accumulation kernel alone; "reduce" = everything after it = "finish" (bucket finish + running sums) +
    // "tree" (over the segment sums) + "combine" (window sums, Horner chain, conversion to affine); "sort" = "prepare" + the sort
    if (detail)
        collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}, {"sort", {0, 3}}, {"accumulate", {3, 4}},
                             {"reduce", {4, 1}}, {"prepare", {0, 6}}, {"finish", {4, 7}}, {"tree", {7, 8}}, {"combine", {8, 1}}});
    else
        collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}, {"sort", {0, 3}},
                             {"accumulate", {3, 4}}, {"reduce", {4, 1}}});
    return rc;
}

// ---- `LinearCombination::lincomb` in its constant-time form (primeorder/src/projective.rs:484-496 -> :532-557; k256
// mul.rs:84-98 -> :112-163): one uniform-schedule multiplication per term (k_var_base_ct: the reference's table, digits and
// additions for that term) and a tree of complete additions over the products, 256 per workgroup and level.  The reference
// interleaves the terms on one accumulator (Straus); the group element is the same and the schedule here depends on n only.
template <class C>
int lincomb_ct_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_xy, const void* d_inf, size_t n, void* d_out_xy,
                   void* d_out_inf) {
    constexpr int NS = Field<C>::NS, WB = WireBytes<C>::value;
    int rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    if (n == 0) {                              // the empty sum
        HIP_TRY(ctx, hipMemsetAsync(d_out_xy, 0, 2 * WB, ctx->stream));
        if (d_out_inf) HIP_TRY(ctx, hipMemsetAsync(d_out_inf, 1, 1, ctx->stream));
        return finish(ctx);
    }
    CtWipe wipe(ctx, WIPE_SCRATCH);
    const size_t tstride = var_base_slots<C>(n);
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ct_flags, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->prefix, ((n + BLOCK - 1) / BLOCK + 1) * 3 * NS * 4)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_var_base_ct<C>(ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n,
                          (uint32_t*)ctx->vtab.p, tstride, (uint32_t*)ctx->proj.p, (uint8_t*)ctx->ct_flags.p, ctx->d_status);
    record(ctx, 3);
    launch_proj_sum<C>(ctx->stream, (uint32_t*)ctx->proj.p, n, (uint32_t*)ctx->prefix.p);
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, 1, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}, {"accumulate", {0, 3}}, {"reduce", {3, 1}}});
    return rc;
}

// ---- compressed points into the path: x + SEC1 tag records are decoded on the device (k_decompress_tagged: one square root
// per point) into the context's scratch, then the ordinary pipeline runs on the x || y records; a record that decodes to no
// point ends the call with ECGPU_ERR_POINT like an off-curve x || y record would
template <class C>
int decode_compressed(ecgpu_ctx* ctx, const void* d_x, const void* d_tag, size_t n) {
    constexpr int WB = WireBytes<C>::value;
    int rc;
    if ((rc = ensure(ctx, ctx->cx_xy, n * 2 * WB + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->cx_inf, n + 16)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    if (n)
        launch_decompress_tagged<C>(ctx->stream, (const uint8_t*)d_x, (const uint8_t*)d_tag, n, (uint8_t*)ctx->cx_xy.p,
                                    (uint8_t*)ctx->cx_inf.p, ctx->d_status);
    return ECGPU_OK;
}
struct KeepStatus {            // the second stage of a two-stage call must not clear the first stage's verdicts
    ecgpu_ctx* ctx;
    explicit KeepStatus(ecgpu_ctx* c) : ctx(c) { ctx->keep_status = true; }
    ~KeepStatus() { ctx->keep_status = false; }
};
template <class C>
int msm_compressed_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_x, const void* d_tag, size_t n, void* d_out_xy,
                       void* d_out_inf) {
    int rc;
    if ((rc = decode_compressed<C>(ctx, d_x, d_tag, n)) != ECGPU_OK) return rc;
    KeepStatus keep(ctx);
    return msm_dev<C>(ctx, d_scalars, ctx->cx_xy.p, ctx->cx_inf.p, n, d_out_xy, d_out_inf);
}
template <class C>
int mul_var_compressed_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_x, const void* d_tag, size_t n, void* d_out_xy,
                           void* d_out_inf) {
    int rc;
    if ((rc = decode_compressed<C>(ctx, d_x, d_tag, n)) != ECGPU_OK) return rc;
    KeepStatus keep(ctx);
    return mul_var_dev<C>(ctx, d_scalars, ctx->cx_xy.p, ctx->cx_inf.p, n, d_out_xy, d_out_inf);
}

// ---- an MSM whose terms are spread over several GPUs: local half / combining half ------------------------
template <class C>
int msm_parts_dev(ecgpu_ctx* ctx, const void* d_scalars, const void* d_xy, const void* d_inf, size_t n, size_t plan_terms,
                  void* d_parts) {
    int rc;
    if (n > msm_max_terms<C>() || plan_terms > msm_max_terms<C>()) {
        ctx->err = "MSM shard of 2^31 (k256: 2^30) or more terms";
        return ECGPU_ERR_ARG;
    }
    if (plan_terms < n) {
        ctx->err = "ecgpu_msm_parts_dev: plan_terms must be at least the shard's term count (and the same on every GPU)";
        return ECGPU_ERR_ARG;
    }
    const int c = ctx->msm_c ? ctx->msm_c : msm_choose_window<C>(plan_terms);
    MsmPlan plan = msm_plan<C>(n, c, msm_use_glv<C>(plan_terms));
    if (ctx->async && ctx->msm_lanes > 1) {
        // Local halves of CONSECUTIVE sharded MSMs on rotating lanes (throughput form): this one runs on its lane's
        // stream and workspace, ordered after what the context's stream holds now (the inputs), beside the exchange and the
        // combining half of the previous one, which the caller keeps on the context's stream:
        //     parts(i) -> lane i % L      ecgpu_msm_parts_join_dev(d_parts(i - 1)); all-gather(i - 1); ecgpu_msm_finish_dev(i - 1)
        // d_parts belongs to the lane until ecgpu_msm_parts_join_dev(d_parts) (the context's stream then waits for it) or
        // ecgpu_synchronize.
        ecgpu_ctx::MsmLane* lp = nullptr;
        if ((rc = next_lane(ctx, plan.workspace_bytes, &lp)) != ECGPU_OK) return rc;
        ecgpu_ctx::MsmLane& l = *lp;
        // A lane remembers ONE record.  If the record of its previous local half was never joined (more local halves in flight than
        // lanes), that half is joined now — the context's stream waits for it before anything queued later —, so that whatever the
        // caller does with the old record afterwards is still ordered behind the kernels that wrote it.
        if (l.parts_out) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, l.ev_done, 0));
        HIP_TRY(ctx, hipEventRecord(l.ev_in, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(l.s, l.ev_in, 0));
        launch_msm_parts<C>(plan, l.s, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n, l.ws.p, (uint32_t*)d_parts,
                            ctx->d_status, l.ev_a, l.ev_b);
        HIP_TRY(ctx, hipEventRecord(l.ev_done, l.s));
        l.parts_out = d_parts;
        ctx->lanes_pending = true;
        return finish(ctx);
    }
    if ((rc = ensure(ctx, ctx->msm_ws, plan.workspace_bytes)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_msm_parts<C>(plan, ctx->stream, (const uint8_t*)d_scalars, (const uint8_t*)d_xy, (const uint8_t*)d_inf, n, ctx->msm_ws.p,
                        (uint32_t*)d_parts, ctx->d_status, ctx->timing_on ? ctx->ev[3] : nullptr, ctx->timing_on ? ctx->ev[4] : nullptr);
    record(ctx, 1);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"total", {0, 1}}, {"sort", {0, 3}}, {"accumulate", {3, 4}}, {"reduce", {4, 1}}});
    return rc;
}

template <class C>
int msm_finish_dev(ecgpu_ctx* ctx, const void* d_parts_all, int nranks, size_t plan_terms, void* d_out_xy, void* d_out_inf) {
    constexpr int NS = Field<C>::NS;
    int rc;
    const int c = ctx->msm_c ? ctx->msm_c : msm_choose_window<C>(plan_terms);
    MsmPlan plan = msm_plan<C>(0, c, msm_use_glv<C>(plan_terms));   // only c, nwin and nparts matter here
    if ((rc = ensure(ctx, ctx->proj, 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->bases, (size_t)plan.nwin * 3 * NS * 4)) != ECGPU_OK) return rc;
    // With local halves in flight on lanes this call does NOT wait for them (that is the point of the lanes): it reads d_parts_all,
    // which the caller's exchange produced on this stream after ecgpu_msm_parts_join_dev, and scratch of its own.
    if (ctx->async && ctx->msm_lanes > 1) {          // (nor does it touch the timing marks: ecgpu_last_timing keeps reading the last lane's)
        launch_msm_finish<C>(plan, ctx->stream, (const uint32_t*)d_parts_all, nranks, (uint32_t*)ctx->bases.p, (uint32_t*)ctx->proj.p,
                             (uint8_t*)d_out_xy, (uint8_t*)d_out_inf);
        return finish(ctx);
    }
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    record(ctx, 0);
    launch_msm_finish<C>(plan, ctx->stream, (const uint32_t*)d_parts_all, nranks, (uint32_t*)ctx->bases.p, (uint32_t*)ctx->proj.p,
                         (uint8_t*)d_out_xy, (uint8_t*)d_out_inf);
    record(ctx, 1);
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}

template <class C>
size_t msm_parts_bytes(const ecgpu_ctx* ctx, size_t plan_terms) {
    return msm_plan<C>(0, ctx->msm_c ? ctx->msm_c : msm_choose_window<C>(plan_terms), msm_use_glv<C>(plan_terms)).parts_bytes;
}

// ---- host-pointer plumbing ---------------------------------------------------------------------------------

int upload(ecgpu_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
    int rc = ensure(ctx, b, bytes ? bytes : 16);
    if (rc != ECGPU_OK) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ECGPU_OK;
}
int download(ecgpu_ctx* ctx, void* host, const DevBuf& b, size_t bytes) {
    if (bytes && host) {
        HIP_TRY(ctx, hipMemcpyAsync(host, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ECGPU_OK;
}

// ---- pipelined host-pointer calls --------------------------------------------------------------------------------------
// A batch of independent units handed over in host memory is cut into chunks of PIPE_CHUNK units: an upload thread and a
// download thread move chunk i + 1 in and chunk i - 1 out (each on its own stream, PCIe is full duplex) while the calling
// thread runs the ordinary device-pointer entry point on chunk i.  Serially the transfers of a 2^20-scalar fixed-base
// batch take five times its compute time.
constexpr size_t PIPE_CHUNK = (size_t)1 << 18;
constexpr size_t PIPE_MIN = (size_t)1 << 19;
constexpr size_t MSM_PIPE_CHUNK = (size_t)1 << 22;   // terms per partial MSM of the host-pointer ecgpu_msm
// ECGPU_MSM_PIPE_LOG2 moves that (tuning knob; the tests use it to drive the chunked path with small inputs); the
// chunked path needs chunk starts that keep p224's 28-byte records 4-byte aligned, which every power of two does
inline size_t msm_pipe_chunk() {
    if (const char* e = knob("ECGPU_MSM_PIPE_LOG2")) {
        int v = atoi(e);
        if (v >= 8 && v <= 26) return (size_t)1 << v;
    }
    return MSM_PIPE_CHUNK;
}

struct PipeIn { const uint8_t* host; DevBuf* dev; size_t unit; };
struct PipeOut { uint8_t* host; DevBuf* dev; size_t unit; };

template <class F>
int pipelined(ecgpu_ctx* ctx, size_t n, const std::vector<PipeIn>& ins, const std::vector<PipeOut>& outs, F&& compute,
              const size_t chunk = PIPE_CHUNK) {
    HIP_TRY(ctx, ctx->up_stream ? hipSuccess : hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, ctx->down_stream ? hipSuccess : hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    int rc;
    for (auto& a : ins) if (a.host && (rc = ensure(ctx, *a.dev, n * a.unit + 16)) != ECGPU_OK) return rc;
    for (auto& o : outs) if ((rc = ensure(ctx, *o.dev, n * o.unit + 16)) != ECGPU_OK) return rc;
    const size_t nchunks = (n + chunk - 1) / chunk;
    std::mutex mu;
    std::condition_variable cv;
    size_t uploaded = 0, computed = 0;
    bool failed = false;
    const int device = ctx->device;
    auto span = [&](size_t i, size_t* off, size_t* m) { *off = i * chunk; *m = n - *off < chunk ? n - *off : chunk; };
    auto up_body = [&] {
        bool ok = hipSetDevice(device) == hipSuccess;
        for (size_t i = 0; i < nchunks; i++) {
            size_t off, m;
            span(i, &off, &m);
            for (auto& a : ins)
                if (ok && a.host)
                    ok = hipMemcpyAsync((uint8_t*)a.dev->p + off * a.unit, a.host + off * a.unit, m * a.unit, hipMemcpyHostToDevice,
                                        ctx->up_stream) == hipSuccess;
            ok = ok && hipStreamSynchronize(ctx->up_stream) == hipSuccess;
            std::lock_guard<std::mutex> g(mu);
            if (!ok) failed = true;
            uploaded = i + 1;
            cv.notify_all();
            if (failed) return;
        }
    };
    auto down_body = [&] {
        bool ok = hipSetDevice(device) == hipSuccess;
        for (size_t i = 0; i < nchunks; i++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return computed > i || failed; });
                if (failed) return;
            }
            size_t off, m;
            span(i, &off, &m);
            for (auto& o : outs)
                if (ok && o.host)
                    ok = hipMemcpyAsync(o.host + off * o.unit, (const uint8_t*)o.dev->p + off * o.unit, m * o.unit, hipMemcpyDeviceToHost,
                                        ctx->down_stream) == hipSuccess;
            ok = ok && hipStreamSynchronize(ctx->down_stream) == hipSuccess;
            if (!ok) {
                std::lock_guard<std::mutex> g(mu);
                failed = true;
                cv.notify_all();
                return;
            }
        }
    };
    std::thread up, down;
    try {                                                   // no C++ exception may cross the C ABI
        up = std::thread(up_body);
        down = std::thread(down_body);
    } catch (...) {
        {
            std::lock_guard<std::mutex> g(mu);
            failed = true;
            cv.notify_all();
        }
        if (up.joinable()) up.join();
        ctx->err = "could not start the transfer threads";
        return ECGPU_ERR_HIP;
    }
    rc = ECGPU_OK;
    for (size_t i = 0; i < nchunks; i++) {
        {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return uploaded > i || failed; });
            if (failed) break;
        }
        size_t off, m;
        span(i, &off, &m);
        int r = compute(off, m);
        std::lock_guard<std::mutex> g(mu);
        if (r != ECGPU_OK) {
            rc = r;
            failed = true;
        }
        computed = i + 1;
        cv.notify_all();
        if (failed) break;
    }
    up.join();
    down.join();
    if (rc == ECGPU_OK && failed) {
        ctx->err = "host <-> device transfer failed";
        rc = ECGPU_ERR_HIP;
    }
    return rc;
}

bool check_ctx(ecgpu_ctx* ctx) {
    if (!ctx) return false;
    if (hipSetDevice(ctx->device) == hipSuccess) return true;
    ctx->err = "hipSetDevice failed";
    return false;
}

// every error return leaves a message for ecgpu_last_error (never a stale one)
int arg_error(ecgpu_ctx* ctx, const char* fn) {
    if (ctx) ctx->err = std::string(fn) + ": NULL, misaligned or inconsistent argument";
    return ECGPU_ERR_ARG;
}
int curve_error(ecgpu_ctx* ctx, const char* fn) {
    if (ctx) ctx->err = std::string(fn) + ": unknown curve id, or the operation does not exist for this curve";
    return ECGPU_ERR_CURVE;
}

}  // namespace

// ================================================================================================================
// C ABI
// ================================================================================================================

namespace {
// shared driver of the two verification shapes: prepare -> a*G + b*Q -> normalise -> compare
enum { VERIFY_ECDSA = 0, VERIFY_SCHNORR = 1, VERIFY_SCHNORR_RAW = 2, VERIFY_SM2DSA = 3, VERIFY_RECOVER = 4, VERIFY_BIGN = 5 };
// mode VERIFY_BIGN: d_h = 32-byte hashes, d_s = 48-byte signatures S0 || S1, d_r unused
// mode VERIFY_SCHNORR_RAW: d_h = messages (msg_len bytes each), d_s = 64-byte signatures, d_q_xy = 32-byte x-only keys
// mode VERIFY_RECOVER (public-key recovery): d_q_xy = the recovery id bytes, d_out_xy receives the keys
template <class C>
int verify_dev(ecgpu_ctx* ctx, int mode, const void* d_h, const void* d_r, const void* d_s, const void* d_q_xy, size_t n,
               int reject_high_s, void* d_ok, size_t msg_len = 0, void* d_out_xy = nullptr) {
    const bool schnorr = mode == VERIFY_SCHNORR || mode == VERIFY_SCHNORR_RAW;
    constexpr int NS = Field<C>::NS;
    const size_t L = 4 * C::N;
    int rc;
    if ((rc = ensure_table<C>(ctx, n)) != ECGPU_OK) return rc;
    if (n == 0) return (int)ECGPU_OK;
    size_t tstride = var_base_slots<C>(n);
    if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_u1, n * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_u2, n * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_q, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_valid, n + 16)) != ECGPU_OK) return rc;
    if (mode != VERIFY_RECOVER && (rc = ensure(ctx, ctx->ec_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_inf, n + 16)) != ECGPU_OK) return rc;
    if (mode == VERIFY_SCHNORR_RAW && (rc = ensure(ctx, ctx->ec_r, n * L)) != ECGPU_OK) return rc;
    // ECDSA verification and recovery invert one scalar per signature: done for the whole batch by Montgomery's trick
    const bool batch_inv = mode == VERIFY_RECOVER || mode == VERIFY_ECDSA;
    if (batch_inv) {
        if ((rc = ensure(ctx, ctx->ec_winv, n * L + 16)) != ECGPU_OK) return rc;
        static_assert(Field<C>::NS >= C::N, "the normalisation's prefix array holds the inverses' prefix products too");
        if ((rc = ensure(ctx, ctx->prefix, n * Field<C>::NS * 4)) != ECGPU_OK) return rc;      // (the size normalize_out asks for later)
    }
    uint32_t* inv_prefix = batch_inv ? (uint32_t*)ctx->prefix.p : nullptr;
    uint8_t* inv_out = batch_inv ? (uint8_t*)ctx->ec_winv.p : nullptr;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    const Table& t = ctx->table[C::ID];
    uint32_t* pa = (uint32_t*)ctx->proj.p;
    uint8_t *u1 = (uint8_t*)ctx->ec_u1.p, *u2 = (uint8_t*)ctx->ec_u2.p, *q = (uint8_t*)ctx->ec_q.p;
    uint8_t* valid = (uint8_t*)ctx->ec_valid.p;
    record(ctx, 0);
    if (mode == VERIFY_SCHNORR_RAW) {
        launch_schnorr_prepare_raw(ctx->stream, (const uint8_t*)d_q_xy, (const uint8_t*)d_h, msg_len, (const uint8_t*)d_s, n, u1,
                                   u2, q, (uint8_t*)ctx->ec_r.p, valid);
        d_r = ctx->ec_r.p;
    } else if (mode == VERIFY_RECOVER)
        launch_ecdsa_recover_prepare<C>(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)d_r, (const uint8_t*)d_s,
                                        (const uint8_t*)d_q_xy, n, reject_high_s, u1, u2, q, valid, inv_prefix, inv_out);
    else if (mode == VERIFY_SM2DSA)
        launch_sm2dsa_prepare<C>(ctx->stream, (const uint8_t*)d_r, (const uint8_t*)d_s, (const uint8_t*)d_q_xy, n, u1, u2, q, valid);
    else if (mode == VERIFY_BIGN)
        launch_bign_prepare(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)d_s, (const uint8_t*)d_q_xy, n, u1, u2, q, valid);
    else if (schnorr)
        launch_schnorr_prepare<C>(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)d_r, (const uint8_t*)d_s,
                                  (const uint8_t*)d_q_xy, n, u1, u2, q, valid);
    else
        launch_ecdsa_prepare<C>(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)d_r, (const uint8_t*)d_s,
                                (const uint8_t*)d_q_xy, n, reject_high_s, u1, u2, q, valid, inv_prefix, inv_out);
    record(ctx, 3);
    launch_fixed_base<C>(ctx->stream, u1, n, (const uint32_t*)t.d, t.w, t.nwin, pa, ctx->d_status);
    launch_var_base<C>(ctx->stream, u2, q, nullptr, n, (uint32_t*)ctx->vtab.p, tstride, nullptr, ctx->d_status, pa);   // pa[i] += u2[i] Q[i]
    record(ctx, 1);
    if ((rc = normalize_out<C>(ctx, n, mode == VERIFY_RECOVER ? d_out_xy : ctx->ec_xy.p, ctx->ec_inf.p)) != ECGPU_OK) return rc;
    if (mode == VERIFY_RECOVER)
        launch_ecdsa_recover_finish<C>(ctx->stream, (uint8_t*)d_out_xy, (const uint8_t*)ctx->ec_inf.p, valid, n, (uint8_t*)d_ok);
    else if (mode == VERIFY_SM2DSA)
        launch_sm2dsa_finish<C>(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)ctx->ec_xy.p, (const uint8_t*)ctx->ec_inf.p,
                                (const uint8_t*)d_r, valid, n, (uint8_t*)d_ok);
    else if (mode == VERIFY_BIGN)
        launch_bign_finish(ctx->stream, (const uint8_t*)d_h, (const uint8_t*)ctx->ec_xy.p, (const uint8_t*)ctx->ec_inf.p,
                           (const uint8_t*)d_s, valid, n, (uint8_t*)d_ok);
    else if (schnorr)
        launch_schnorr_finish<C>(ctx->stream, (const uint8_t*)ctx->ec_xy.p, (const uint8_t*)ctx->ec_inf.p, (const uint8_t*)d_r,
                                 valid, n, (uint8_t*)d_ok);
    else
        launch_ecdsa_finish<C>(ctx->stream, (const uint8_t*)ctx->ec_xy.p, (const uint8_t*)ctx->ec_inf.p, (const uint8_t*)d_r,
                               valid, n, (uint8_t*)d_ok);
    record(ctx, 2);
    rc = finish(ctx);
    collect_timing(ctx, {{"recode", {0, 3}}, {"main", {3, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
    return rc;
}
}  // namespace

extern "C" {

const char* ecgpu_version(void) { return "ecgpu 0.1 (gfx950)"; }

size_t ecgpu_field_bytes(int curve) {
    switch (curve) {
    case ECGPU_K256: case ECGPU_P256: return 32;
    case ECGPU_P384: case ECGPU_BP384: case ECGPU_BP384T1: return 48;
    case ECGPU_SM2: case ECGPU_BP256: case ECGPU_BP256T1: case ECGPU_BIGN256: return 32;
    case ECGPU_P224: return 28;
    case ECGPU_P192: return 24;
    case ECGPU_P521: return 66;
    default: return 0;
    }
}

int ecgpu_device_count(void) {
    int count = 0, usable = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    for (int d = 0; d < count; d++) {          // the leading run of gfx950 devices: a group lists devices 0 .. count - 1
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;
        usable++;
    }
    return usable;
}

int ecgpu_init(ecgpu_ctx** out, int device) {
    if (!out) return ECGPU_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ECGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ECGPU_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ECGPU_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "ecgpu: device %d is %s; this library is built for gfx950 only\n", device, prop.gcnArchName);
        return ECGPU_ERR_NO_DEVICE;
    }
    ecgpu_ctx* ctx = new (std::nothrow) ecgpu_ctx();
    if (!ctx) return ECGPU_ERR_OOM;
    ctx->device = device;
    bool ok = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
    ctx->stream = ctx->own_stream;
    ok = ok && hipMalloc(reinterpret_cast<void**>(&ctx->d_status), sizeof(int)) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_status), sizeof(int)) == hipSuccess;
    for (auto& e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        ecgpu_destroy(ctx);
        return ECGPU_ERR_HIP;
    }
    *out = ctx;
    return ECGPU_OK;
}

void ecgpu_destroy(ecgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf* b : {&ctx->proj, &ctx->prefix, &ctx->vtab, &ctx->bases, &ctx->in0, &ctx->in1, &ctx->in2, &ctx->in3,
                      &ctx->out0, &ctx->out1, &ctx->msm_ws, &ctx->ec_u1, &ctx->ec_u2, &ctx->ec_winv, &ctx->ec_q, &ctx->ec_valid, &ctx->ec_xy,
                      &ctx->ec_inf, &ctx->ec_r, &ctx->ec_e, &ctx->ec_s, &ctx->ec_id})
        if (b->p) (void)hipFree(b->p);
    for (DevBuf* b : {&ctx->ct_flags, &ctx->cx_xy, &ctx->cx_inf})
        if (b->p) (void)hipFree(b->p);
    for (auto& l : ctx->lane) {
        if (l.s) (void)hipStreamSynchronize(l.s);
        for (DevBuf* b : {&l.ws, &l.proj, &l.prefix})
            if (b->p) (void)hipFree(b->p);
        for (hipEvent_t e : {l.ev_in, l.ev_a, l.ev_b, l.ev_done})
            if (e) (void)hipEventDestroy(e);
        if (l.s) (void)hipStreamDestroy(l.s);
    }
    for (int id = 0; id < 12; id++) {                            // the last context of the device frees the shared tables
        release_table(ctx, id);
        release_ct_lut(ctx, id);
    }
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
    if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
    delete ctx;
}

const char* ecgpu_last_error(const ecgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void* ecgpu_host_alloc(ecgpu_ctx* ctx, size_t bytes) {
    if (!check_ctx(ctx) || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        ctx->err = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void ecgpu_host_free(ecgpu_ctx* ctx, void* p) {
    if (p && check_ctx(ctx)) (void)hipHostFree(p);
}

void* ecgpu_dev_alloc(ecgpu_ctx* ctx, size_t bytes) {
    if (!check_ctx(ctx) || bytes == 0) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        ctx->err = "hipMalloc failed";
        return nullptr;
    }
    return p;
}

void ecgpu_dev_free(ecgpu_ctx* ctx, void* d_ptr) {
    if (!d_ptr || !check_ctx(ctx)) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}

int ecgpu_copy_to_device(ecgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (bytes == 0) return ECGPU_OK;
    if (!d_dst || !h_src) {
        ctx->err = "ecgpu_copy_to_device: null pointer";
        return arg_error(ctx, __func__);
    }
    if (int rc = join_lanes(ctx)) return rc;             // (an MSM on a lane may be writing / reading the buffer)
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ECGPU_OK;
}

int ecgpu_copy_to_host(ecgpu_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (bytes == 0) return ECGPU_OK;
    if (!h_dst || !d_src) {
        ctx->err = "ecgpu_copy_to_host: null pointer";
        return arg_error(ctx, __func__);
    }
    if (int rc = join_lanes(ctx)) return rc;             // (an MSM on a lane may be writing / reading the buffer)
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ECGPU_OK;
}

int ecgpu_set_stream(ecgpu_ctx* ctx, void* stream) {
    if (!ctx) return arg_error(ctx, __func__);
    if (ctx->async && ctx->pending && check_ctx(ctx)) (void)drain(ctx);     // the status word follows the old stream
    ctx->stream = stream ? reinterpret_cast<hipStream_t>(stream) : ctx->own_stream;
    return ECGPU_OK;
}

int ecgpu_set_base_window(ecgpu_ctx* ctx, int curve, int window_bits) {
    if (!ctx || curve < 0 || curve > 11) return ECGPU_ERR_CURVE;
    if (window_bits == 0) {                                   // back to the automatic choice below the curve's default maximum
        ctx->want_w[curve] = ecgpu_ctx().want_w[curve];
        ctx->w_pinned[curve] = false;
        return ECGPU_OK;
    }
    if (window_bits < 4 || window_bits > 26) return arg_error(ctx, __func__);
    ctx->want_w[curve] = window_bits;
    ctx->w_pinned[curve] = true;
    return ECGPU_OK;
}

int ecgpu_set_table_policy(ecgpu_ctx* ctx, int policy) {
    if (!ctx || (policy != ECGPU_TABLE_ADAPTIVE && policy != ECGPU_TABLE_EAGER)) return arg_error(ctx, __func__);
    ctx->table_policy = policy;
    return ECGPU_OK;
}

int ecgpu_set_table_budget(ecgpu_ctx* ctx, size_t max_table_bytes) {
    if (!ctx) return arg_error(ctx, __func__);
    ctx->table_budget = max_table_bytes;
    return ECGPU_OK;
}

int ecgpu_base_table_info(ecgpu_ctx* ctx, int curve, int* window_bits, size_t* table_bytes, double* build_ms) {
    if (!ctx || curve < 0 || curve > 11) return ECGPU_ERR_CURVE;
    const Table& t = ctx->table[curve];
    int w = 0;
    size_t bytes = 0;
    double ms = 0;
    if (t.d) {
        TableRegistry& reg = table_registry(ctx->device);
        std::lock_guard<std::mutex> lock(reg.mu);
        auto it = reg.tabs.find(std::make_tuple(ctx->device, curve, t.w));
        if (it != reg.tabs.end()) {
            w = t.w;
            bytes = it->second.bytes;
            ms = it->second.build_ms;
        }
    }
    if (window_bits) *window_bits = w;
    if (table_bytes) *table_bytes = bytes;
    if (build_ms) *build_ms = ms;
    return ECGPU_OK;
}

int ecgpu_set_msm_window(ecgpu_ctx* ctx, int window_bits) {
    if (!ctx) return arg_error(ctx, __func__);
    if (window_bits != 0 && (window_bits < 4 || window_bits > 16)) return arg_error(ctx, __func__);
    ctx->msm_c = window_bits;
    return ECGPU_OK;
}

int ecgpu_set_msm_lanes(ecgpu_ctx* ctx, int lanes) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (lanes < 1 || lanes > 4) return arg_error(ctx, __func__);
    int rc = ctx->async ? drain(ctx) : ECGPU_OK;       // nothing in flight on a lane while the mode changes
    ctx->msm_lanes = lanes;
    return rc;
}

int ecgpu_set_timing(ecgpu_ctx* ctx, int on) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    ctx->timing_on = on != 0;
    if (!ctx->timing_on) {
        ctx->spans.clear();
        ctx->timing.clear();
    }
    return ECGPU_OK;
}

int ecgpu_set_async(ecgpu_ctx* ctx, int on) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    int rc = ecgpu_synchronize(ctx);          // what was queued so far is reported here
    if (on && !ctx->async) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
        ctx->deferred = 0;
    }
    ctx->async = on != 0;
    return rc;
}

int ecgpu_synchronize(ecgpu_ctx* ctx) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!ctx->async) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return ECGPU_OK;
    }
    // (nothing queued since the last drain: the status word on the device is clear and there is nothing to wait for — a caller that
    // switches the mode around every call, like ecgpu_group_msm_dev, pays for one round trip per call, not two)
    if (ctx->pending || ctx->lanes_pending) {
        int rc = drain(ctx);
        if (rc != ECGPU_OK) return rc;
    } else {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int st = ctx->deferred;
    ctx->deferred = 0;
    return status_error(ctx, st);
}

int ecgpu_wipe(ecgpu_ctx* ctx) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    for (auto& l : ctx->lane)
        if (l.s) HIP_TRY(ctx, hipStreamSynchronize(l.s));
    for (DevBuf* b : {&ctx->proj, &ctx->prefix, &ctx->vtab, &ctx->bases, &ctx->in0, &ctx->in1, &ctx->in2, &ctx->in3, &ctx->out0, &ctx->out1,
                      &ctx->msm_ws, &ctx->ec_u1, &ctx->ec_u2, &ctx->ec_winv, &ctx->ec_q, &ctx->ec_valid, &ctx->ec_xy, &ctx->ec_inf, &ctx->ec_r, &ctx->ec_e,
                      &ctx->ec_s, &ctx->ec_id, &ctx->ct_flags, &ctx->cx_xy, &ctx->cx_inf})
        if (b->p) HIP_TRY(ctx, hipMemsetAsync(b->p, 0, b->cap, ctx->stream));
    for (auto& l : ctx->lane)
        for (DevBuf* b : {&l.ws, &l.proj, &l.prefix})
            if (b->p) HIP_TRY(ctx, hipMemsetAsync(b->p, 0, b->cap, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ECGPU_OK;
}

// test-only (not in include/ecgpu.h): comb tables above `mb` MiB are refused as if the allocation had failed; 0 switches it off
void ecgpu_testhook_table_max_mb(size_t mb) {
    g_test_table_max_mb.store(mb);
    for (int dev = 0; dev < 64; dev++) {                          // what the hook made the registry remember goes with it
        TableRegistry& reg = table_registry(dev);
        std::lock_guard<std::mutex> lock(reg.mu);
        reg.nofit.clear();
    }
}

int ecgpu_last_timing(const ecgpu_ctx* ctx_in, const char* name, double* ms) {
    if (!ctx_in || !name || !ms) return ECGPU_ERR_ARG;
    ecgpu_ctx* ctx = const_cast<ecgpu_ctx*>(ctx_in);
    if (ctx->lane_last >= 0 && ctx->timing.empty() && ctx->spans.empty() && std::string(name) == "accumulate") {
        // the last MSM went to a lane: the duration of its accumulation kernel with whatever ran beside it
        ecgpu_ctx::MsmLane& l = ctx->lane[ctx->lane_last];
        float t = 0;
        if (!check_ctx(ctx) || !l.s || hipStreamSynchronize(l.s) != hipSuccess || hipEventElapsedTime(&t, l.ev_a, l.ev_b) != hipSuccess)
            return ECGPU_ERR_HIP;
        *ms = t;
        return ECGPU_OK;
    }
    if (!ctx->spans.empty()) {                 // asynchronous call: its events are complete once the stream has drained
        if (!check_ctx(ctx) || hipStreamSynchronize(ctx->stream) != hipSuccess) return ECGPU_ERR_HIP;
        resolve_timing(ctx);
    }
    auto it = ctx->timing.find(name);
    if (it == ctx->timing.end()) return ECGPU_ERR_ARG;
    *ms = it->second;
    return ECGPU_OK;
}

// ---- device-pointer entry points ----

int ecgpu_batch_mul_base_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, size_t n, void* d_out_xy,
                             void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_out_xy || !aligned16(d_scalars) || !aligned16(d_out_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) { return mul_base_dev<decltype(c)>(ctx, d_scalars, n, d_out_xy, d_out_inf); });
}

int ecgpu_batch_mul_base_compressed_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, size_t n, void* d_out_x,
                                        void* d_out_tag) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_out_x || !d_out_tag || !aligned16(d_scalars) || !aligned16(d_out_x))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) { return mul_base_dev<decltype(c)>(ctx, d_scalars, n, d_out_x, d_out_tag, true); });
}

int ecgpu_batch_mul_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy,
                        const void* d_points_inf, size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_points_xy || !d_out_xy || !aligned16(d_scalars) || !aligned16(d_points_xy) ||
              !aligned16(d_out_xy)))
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return mul_var_dev<decltype(c)>(ctx, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_batch_mul_base_ct_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_out_xy || !aligned16(d_scalars) || !aligned16(d_out_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) { return mul_base_ct_dev<decltype(c)>(ctx, d_scalars, n, d_out_xy, d_out_inf); });
}

int ecgpu_batch_mul_ct_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, const void* d_points_inf,
                           size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_points_xy || !d_out_xy || !aligned16(d_scalars) || !aligned16(d_points_xy) ||
              !aligned16(d_out_xy)))
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return mul_var_ct_dev<decltype(c)>(ctx, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_msm_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, const void* d_points_inf,
                  size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_out_xy || !aligned16(d_out_xy)) return arg_error(ctx, __func__);
    if (n && (!d_scalars || !d_points_xy || !aligned16(d_scalars) || !aligned16(d_points_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return msm_dev<decltype(c)>(ctx, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_lincomb_ct_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, const void* d_points_inf,
                         size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_out_xy || !aligned16(d_out_xy)) return arg_error(ctx, __func__);
    if (n && (!d_scalars || !d_points_xy || !aligned16(d_scalars) || !aligned16(d_points_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return lincomb_ct_dev<decltype(c)>(ctx, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_msm_compressed_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_x, const void* d_points_tag,
                             size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_out_xy || !aligned16(d_out_xy)) return arg_error(ctx, __func__);
    if (n && (!d_scalars || !d_points_x || !d_points_tag || !aligned16(d_scalars) || !aligned16(d_points_x)))
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return msm_compressed_dev<decltype(c)>(ctx, d_scalars, d_points_x, d_points_tag, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_batch_mul_compressed_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_x,
                                   const void* d_points_tag, size_t n, void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_scalars || !d_points_x || !d_points_tag || !d_out_xy || !aligned16(d_scalars) || !aligned16(d_points_x) ||
              !aligned16(d_out_xy)))
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return mul_var_compressed_dev<decltype(c)>(ctx, d_scalars, d_points_x, d_points_tag, n, d_out_xy, d_out_inf);
    });
}

size_t ecgpu_msm_parts_bytes(ecgpu_ctx* ctx, int curve, size_t plan_terms) {
    if (!check_ctx(ctx)) return 0;
    size_t bytes = 0;
    (void)dispatch(curve, [&](auto c) {
        bytes = msm_parts_bytes<decltype(c)>(ctx, plan_terms);
        return (int)ECGPU_OK;
    });
    return bytes;
}

int ecgpu_msm_plan_window(ecgpu_ctx* ctx, int curve, size_t plan_terms) {
    if (!check_ctx(ctx)) return 0;
    int c = 0;
    (void)dispatch(curve, [&](auto cv) {
        c = ctx->msm_c ? ctx->msm_c : msm_choose_window<decltype(cv)>(plan_terms);
        return (int)ECGPU_OK;
    });
    return c;
}

int ecgpu_msm_parts_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, const void* d_points_inf,
                        size_t n, size_t plan_terms, void* d_parts) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_parts || !aligned16(d_parts)) return arg_error(ctx, __func__);
    if (n && (!d_scalars || !d_points_xy || !aligned16(d_scalars) || !aligned16(d_points_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return msm_parts_dev<decltype(c)>(ctx, d_scalars, d_points_xy, d_points_inf, n, plan_terms, d_parts);
    });
}

int ecgpu_msm_parts_join_dev(ecgpu_ctx* ctx, const void* d_parts) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_parts) return arg_error(ctx, __func__);
    for (auto& l : ctx->lane)
        if (l.s && l.ev_done && l.parts_out == d_parts) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, l.ev_done, 0));
            l.parts_out = nullptr;
        }
    return ECGPU_OK;          // (written on the context's own stream, or already joined: nothing to wait for)
}

int ecgpu_msm_finish_dev(ecgpu_ctx* ctx, int curve, const void* d_parts_all, int nranks, size_t plan_terms, void* d_out_xy,
                         void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_parts_all || !aligned16(d_parts_all) || !d_out_xy || !aligned16(d_out_xy) || nranks < 1 || nranks > 4096)
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return msm_finish_dev<decltype(c)>(ctx, d_parts_all, nranks, plan_terms, d_out_xy, d_out_inf);
    });
}

int ecgpu_batch_normalize_dev(ecgpu_ctx* ctx, int curve, const void* d_points_xyz, size_t n, void* d_out_xy,
                              void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_points_xyz || !d_out_xy || !aligned16(d_points_xyz) || !aligned16(d_out_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) { return normalize_dev<decltype(c)>(ctx, d_points_xyz, n, d_out_xy, d_out_inf); });
}

int ecgpu_point_sum_dev(ecgpu_ctx* ctx, int curve, const void* d_points_xy, const void* d_points_inf, size_t n,
                        void* d_out_xy, void* d_out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (!d_out_xy || !aligned16(d_out_xy) || (n && (!d_points_xy || !aligned16(d_points_xy)))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return point_sum_dev<decltype(c)>(ctx, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf);
    });
}

int ecgpu_batch_mul_base_and_mul_add_dev(ecgpu_ctx* ctx, int curve, const void* d_a, const void* d_b,
                                         const void* d_points_xy, const void* d_points_inf, size_t n, void* d_out_xy,
                                         void* d_out_inf) {
    // aG + bP per element = two-term lincomb (mul_backend.rs:29-40).  Evaluated as a*G (table kernel) and
    // b*P (variable-base kernel) into projective scratch halves, then one complete addition per element.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_a || !d_b || !d_points_xy || !d_out_xy || !aligned16(d_a) || !aligned16(d_b) ||
              !aligned16(d_points_xy) || !aligned16(d_out_xy)))
        return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        using C = decltype(c);
        constexpr int N = C::N, NS = Field<C>::NS;
    (void)N;
        int rc;
        if ((rc = ensure_table<C>(ctx, n)) != ECGPU_OK) return rc;
        if (n == 0) return (int)ECGPU_OK;
        size_t tstride = var_base_slots<C>(n);
        if ((rc = ensure(ctx, ctx->proj, n * 3 * NS * 4)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->vtab, tstride * var_base_tab_words<C>() * 4)) != ECGPU_OK) return rc;
        if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
        const Table& t = ctx->table[C::ID];
        uint32_t* pa = (uint32_t*)ctx->proj.p;
            record(ctx, 0);
        launch_fixed_base<C>(ctx->stream, (const uint8_t*)d_a, n, (const uint32_t*)t.d, t.w, t.nwin, pa, ctx->d_status);
        launch_var_base<C>(ctx->stream, (const uint8_t*)d_b, (const uint8_t*)d_points_xy, (const uint8_t*)d_points_inf, n,
                           (uint32_t*)ctx->vtab.p, tstride, nullptr, ctx->d_status, pa);                                   // pa[i] += b[i] P[i]
        record(ctx, 1);
        if ((rc = normalize_out<C>(ctx, n, d_out_xy, d_out_inf)) != ECGPU_OK) return rc;
        record(ctx, 2);
        rc = finish(ctx);
        collect_timing(ctx, {{"main", {0, 1}}, {"normalize", {1, 2}}, {"total", {0, 2}}});
        return rc;
    });
}

int ecgpu_ecdsa_verify_batch_dev(ecgpu_ctx* ctx, int curve, const void* d_z, const void* d_r, const void* d_s,
                                 const void* d_q_xy, size_t n, int reject_high_s, void* d_ok) {
    // per element: u1 = z/s, u2 = r/s (mod n), R = u1 G + u2 Q (the kernels of ecgpu_batch_mul_base_and_mul_add),
    // ok = x(R) mod n == r.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_z || !d_r || !d_s || !d_q_xy || !d_ok || !aligned16(d_z) || !aligned16(d_r) || !aligned16(d_s) ||
              !aligned16(d_q_xy)))
        return arg_error(ctx, __func__);
    if (curve == ECGPU_SM2 || curve == ECGPU_BIGN256)          // sm2 signatures are SM2DSA (sm2/src/dsa.rs), bign's its own scheme
        return curve_error(ctx, __func__);                     // (bignp256/src/ecdsa.rs) — not ECDSA
    return dispatch(curve, [&](auto c) {
        return verify_dev<decltype(c)>(ctx, VERIFY_ECDSA, d_z, d_r, d_s, d_q_xy, n, reject_high_s, d_ok);
    });
}

int ecgpu_ecdsa_verify_msg_batch_dev(ecgpu_ctx* ctx, int curve, const void* d_q_xy, const void* d_msgs, size_t msg_len,
                                     const void* d_sigs, size_t n, int reject_high_s, void* d_ok) {
    // Verifier::verify(msg, sig): the curve's digest on the device, z = bits2field(digest), then the prehash path.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_q_xy || !d_sigs || !d_ok || (msg_len && !d_msgs) || !aligned16(d_q_xy) || !aligned16(d_sigs))) return arg_error(ctx, __func__);
    if (curve == ECGPU_SM2 || curve == ECGPU_BIGN256 || curve == ECGPU_P192)   // not ECDSA curves; p192 has no `DigestAlgorithm` (p192/src/ecdsa.rs)
        return curve_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        using C = decltype(c);
        if (n == 0) return (int)ECGPU_OK;
        const size_t L = WireBytes<C>::value;
        int rc;
        if ((rc = ensure(ctx, ctx->ec_e, n * L + 16)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->ec_r, n * L + 16)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->ec_s, n * L + 16)) != ECGPU_OK) return rc;
        launch_ecdsa_hash_msg<C>(ctx->stream, (const uint8_t*)d_msgs, msg_len, (const uint8_t*)d_sigs, n, (uint8_t*)ctx->ec_e.p,
                                 (uint8_t*)ctx->ec_r.p, (uint8_t*)ctx->ec_s.p);
        return verify_dev<C>(ctx, VERIFY_ECDSA, ctx->ec_e.p, ctx->ec_r.p, ctx->ec_s.p, d_q_xy, n, reject_high_s, d_ok);
    });
}

int ecgpu_ecdsa_recover_batch_dev(ecgpu_ctx* ctx, int curve, const void* d_z, const void* d_r, const void* d_s,
                                  const void* d_recid, size_t n, int reject_high_s, void* d_out_xy, void* d_ok) {
    // per element: R = decompress(r or r + n, parity), key = -(z/r) G + (s/r) R.  See ecgpu_ecdsa.h / ecgpu_verify.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_z || !d_r || !d_s || !d_recid || !d_out_xy || !d_ok || !aligned16(d_z) || !aligned16(d_r) || !aligned16(d_s) ||
              !aligned16(d_out_xy)))
        return arg_error(ctx, __func__);
    if (curve == ECGPU_SM2 || curve == ECGPU_BIGN256)   // not ECDSA curves
        return curve_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        return verify_dev<decltype(c)>(ctx, VERIFY_RECOVER, d_z, d_r, d_s, d_recid, n, reject_high_s, d_ok, 0, d_out_xy);
    });
}

int ecgpu_sm2dsa_verify_batch_dev(ecgpu_ctx* ctx, const void* d_e, const void* d_r, const void* d_s, const void* d_q_xy, size_t n,
                                  void* d_ok) {
    // SM2DSA on the prehash: t = r + s, (x1, y1) = s G + t Q, ok = (e + x1 mod n == r).  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_e || !d_r || !d_s || !d_q_xy || !d_ok || !aligned16(d_e) || !aligned16(d_r) || !aligned16(d_s) ||
              !aligned16(d_q_xy)))
        return arg_error(ctx, __func__);
    return verify_dev<Sm2Params>(ctx, VERIFY_SM2DSA, d_e, d_r, d_s, d_q_xy, n, 0, d_ok);
}

int ecgpu_sm2dsa_verify_msg_batch_dev(ecgpu_ctx* ctx, const void* d_distid, size_t distid_len, const void* d_q_xy, const void* d_msgs,
                                      size_t msg_len, const void* d_sigs, size_t n, void* d_ok) {
    // VerifyingKey::new(distid, Q)?.verify(msg, sig): Z and e = SM3(Z || M) on the device, then the prehash path.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (distid_len > 8191 || (n && (!d_q_xy || !d_sigs || !d_ok || (msg_len && !d_msgs) || (distid_len && !d_distid) || !aligned16(d_q_xy) ||
                                    !aligned16(d_sigs))))
        return arg_error(ctx, __func__);
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->ec_e, n * 32)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_r, n * 32)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_s, n * 32)) != ECGPU_OK) return rc;
    launch_sm2dsa_hash_msg(ctx->stream, (const uint8_t*)d_distid, distid_len, (const uint8_t*)d_q_xy, (const uint8_t*)d_msgs, msg_len,
                           (const uint8_t*)d_sigs, n, (uint8_t*)ctx->ec_e.p, (uint8_t*)ctx->ec_r.p, (uint8_t*)ctx->ec_s.p);
    return verify_dev<Sm2Params>(ctx, VERIFY_SM2DSA, ctx->ec_e.p, ctx->ec_r.p, ctx->ec_s.p, d_q_xy, n, 0, d_ok);
}

int ecgpu_bign_verify_batch_dev(ecgpu_ctx* ctx, const void* d_h, const void* d_sigs, const void* d_q_xy, size_t n, void* d_ok) {
    // bign on the prehash: R = ((S1 + H) mod q) G + (S0 + 2^128) Q, ok = (S0 == belt-hash(OID || x(R) || H)[..16]).  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_h || !d_sigs || !d_q_xy || !d_ok || !aligned16(d_h) || !aligned16(d_sigs) || !aligned16(d_q_xy)))
        return arg_error(ctx, __func__);
    return verify_dev<Bign256Params>(ctx, VERIFY_BIGN, d_h, nullptr, d_sigs, d_q_xy, n, 0, d_ok);
}

int ecgpu_bign_verify_msg_batch_dev(ecgpu_ctx* ctx, const void* d_q_xy, const void* d_msgs, size_t msg_len, const void* d_sigs, size_t n,
                                    void* d_ok) {
    // VerifyingKey::verify(msg, sig): H = belt-hash(msg) on the device, then the prehash path.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_q_xy || !d_sigs || !d_ok || (msg_len && !d_msgs) || !aligned16(d_q_xy) || !aligned16(d_sigs)))
        return arg_error(ctx, __func__);
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->ec_e, n * 32)) != ECGPU_OK) return rc;
    launch_bign_hash_msg(ctx->stream, (const uint8_t*)d_msgs, msg_len, n, (uint8_t*)ctx->ec_e.p);
    return verify_dev<Bign256Params>(ctx, VERIFY_BIGN, ctx->ec_e.p, nullptr, d_sigs, d_q_xy, n, 0, d_ok);
}

int ecgpu_schnorr_verify_batch_dev(ecgpu_ctx* ctx, const void* d_e, const void* d_r, const void* d_s, const void* d_p_xy,
                                   size_t n, void* d_ok) {
    // BIP340 over secp256k1: R = s G - e P, ok = R finite, y(R) even, x(R) == r.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_e || !d_r || !d_s || !d_p_xy || !d_ok || !aligned16(d_e) || !aligned16(d_r) || !aligned16(d_s) ||
              !aligned16(d_p_xy)))
        return arg_error(ctx, __func__);
    return verify_dev<K256Params>(ctx, VERIFY_SCHNORR, d_e, d_r, d_s, d_p_xy, n, 0, d_ok);
}

int ecgpu_schnorr_verify_raw_batch_dev(ecgpu_ctx* ctx, const void* d_pk_x, const void* d_msgs, size_t msg_len, const void* d_sigs,
                                       size_t n, void* d_ok) {
    // VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig) from wire bytes: lift_x, challenge hash, s G - e P.  See ecgpu_ecdsa.h.
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_pk_x || !d_sigs || !d_ok || (msg_len && !d_msgs) || !aligned16(d_pk_x) || !aligned16(d_sigs))) return arg_error(ctx, __func__);
    return verify_dev<K256Params>(ctx, VERIFY_SCHNORR_RAW, d_msgs, nullptr, d_sigs, d_pk_x, n, 0, d_ok, msg_len);
}

static int ecdh_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_x, void* d_ok,
                    bool ct) {
    // SharedSecret_i = x(k_i * P_i): the variable-base kernel (ct: its uniform-schedule form), normalisation into scratch,
    // x extraction
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_out_x || !d_ok || !aligned16(d_out_x))) return arg_error(ctx, __func__);
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    int rc;
    CtWipe wipe(ctx, ct ? WIPE_EC : 0);             // (declared first: runs after the x extraction below has been queued)
    if ((rc = ensure(ctx, ctx->ec_xy, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->ec_inf, n + 16)) != ECGPU_OK) return rc;
    rc = ct ? ecgpu_batch_mul_ct_dev(ctx, curve, d_scalars, d_points_xy, nullptr, n, ctx->ec_xy.p, ctx->ec_inf.p)
            : ecgpu_batch_mul_dev(ctx, curve, d_scalars, d_points_xy, nullptr, n, ctx->ec_xy.p, ctx->ec_inf.p);
    if (rc != ECGPU_OK) return rc;
    if (n == 0) return ECGPU_OK;
    return dispatch(curve, [&](auto c) -> int {
        using C = decltype(c);
        launch_extract_x<C>(ctx->stream, (const uint8_t*)ctx->ec_xy.p, (const uint8_t*)ctx->ec_inf.p, n, (uint8_t*)d_out_x,
                            (uint8_t*)d_ok);
        HIP_TRY(ctx, hipGetLastError());
        if (!ctx->async) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return (int)ECGPU_OK;
    });
}
int ecgpu_batch_ecdh_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_x,
                         void* d_ok) {
    return ecdh_dev(ctx, curve, d_scalars, d_points_xy, n, d_out_x, d_ok, false);
}
int ecgpu_batch_ecdh_ct_dev(ecgpu_ctx* ctx, int curve, const void* d_scalars, const void* d_points_xy, size_t n, void* d_out_x,
                            void* d_ok) {
    return ecdh_dev(ctx, curve, d_scalars, d_points_xy, n, d_out_x, d_ok, true);
}

int ecgpu_batch_decompress_dev(ecgpu_ctx* ctx, int curve, const void* d_xs, const void* d_y_is_odd, size_t n, void* d_out_xy,
                               void* d_ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    if (n && (!d_xs || !d_y_is_odd || !d_out_xy || !d_ok || !aligned16(d_xs) || !aligned16(d_out_xy))) return arg_error(ctx, __func__);
    return dispatch(curve, [&](auto c) {
        using C = decltype(c);
        if (n == 0) return (int)ECGPU_OK;
        int rc;
        if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
        record(ctx, 0);
        launch_decompress<C>(ctx->stream, (const uint8_t*)d_xs, (const uint8_t*)d_y_is_odd, n, (uint8_t*)d_out_xy, (uint8_t*)d_ok);
        record(ctx, 1);
        rc = finish(ctx);
        collect_timing(ctx, {{"main", {0, 1}}, {"total", {0, 1}}});
        return rc;
    });
}

// ---- host-pointer entry points ----

// (ct: the uniform-schedule form; the host-side plumbing is the same)
static int batch_mul_base_host(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, size_t n, uint8_t* out_xy, uint8_t* out_inf,
                               bool ct, const char* fn) {
    const auto dev = ct ? ecgpu_batch_mul_base_ct_dev : ecgpu_batch_mul_base_dev;
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, fn);
    if (n && (!scalars || !out_xy)) return arg_error(ctx, fn);
    CtWipe wipe(ctx, ct ? WIPE_STAGING : 0);          // (after everything below, before SyncScope restores the mode)
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{scalars, &ctx->in0, L}}, {{out_xy, &ctx->out0, 2 * L}, {out_inf, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, m,
                                                             (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = dev(ctx, curve, ctx->in0.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_batch_mul_base(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    return batch_mul_base_host(ctx, curve, scalars, n, out_xy, out_inf, false, __func__);
}
int ecgpu_batch_mul_base_ct(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    return batch_mul_base_host(ctx, curve, scalars, n, out_xy, out_inf, true, __func__);
}

int ecgpu_batch_mul_base_compressed(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, size_t n, uint8_t* out_x,
                                    uint8_t* out_tag) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!scalars || !out_x || !out_tag)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{scalars, &ctx->in0, L}}, {{out_x, &ctx->out0, L}, {out_tag, &ctx->out1, 1}}, [&](size_t off, size_t m) {
            return ecgpu_batch_mul_base_compressed_dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, m, (uint8_t*)ctx->out0.p + off * L,
                                                       (uint8_t*)ctx->out1.p + off);
        });
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_batch_mul_base_compressed_dev(ctx, curve, ctx->in0.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_x, ctx->out0, n * L)) != ECGPU_OK) return rc;
    return download(ctx, out_tag, ctx->out1, n);
}

static int batch_mul_host(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf,
                          size_t n, uint8_t* out_xy, uint8_t* out_inf, bool ct, const char* fn) {
    const auto dev = ct ? ecgpu_batch_mul_ct_dev : ecgpu_batch_mul_dev;
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, fn);
    if (n && (!scalars || !points_xy || !out_xy)) return arg_error(ctx, fn);
    CtWipe wipe(ctx, ct ? WIPE_STAGING : 0);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{scalars, &ctx->in0, L}, {points_xy, &ctx->in1, 2 * L}, {points_inf, &ctx->in2, 1}},
                         {{out_xy, &ctx->out0, 2 * L}, {out_inf, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L,
                                                        points_inf ? (uint8_t*)ctx->in2.p + off : nullptr, m,
                                                        (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (points_inf && (rc = upload(ctx, ctx->in2, points_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = dev(ctx, curve, ctx->in0.p, ctx->in1.p, points_inf ? ctx->in2.p : nullptr, n, ctx->out0.p,
                                  ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_batch_mul(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf, size_t n,
                    uint8_t* out_xy, uint8_t* out_inf) {
    return batch_mul_host(ctx, curve, scalars, points_xy, points_inf, n, out_xy, out_inf, false, __func__);
}
int ecgpu_batch_mul_ct(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf, size_t n,
                       uint8_t* out_xy, uint8_t* out_inf) {
    return batch_mul_host(ctx, curve, scalars, points_xy, points_inf, n, out_xy, out_inf, true, __func__);
}

int ecgpu_msm(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf,
              size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (!out_xy || (n && (!scalars || !points_xy))) return arg_error(ctx, __func__);
    int rc;
    const size_t pipe_chunk = msm_pipe_chunk();
    if (n >= 2 * pipe_chunk) {
        // An MSM needs all of its terms before its sort can start, and 96 (144) bytes per term take longer to upload than
        // the MSM takes to compute: sum_i k_i P_i is computed as one MSM per chunk of 2^22 terms, each under the upload of
        // the next chunk, and the partial sums are added at the end.  The partial records sit at a pitch of 2L bytes
        // (56 for p224, 132 for p521: not 16-byte multiples), so the internal implementations are called directly — the
        // kernels of those curves use 4-byte / byte accesses; only the public *_dev entry points insist on 16-byte bases.
        const size_t nparts = (n + pipe_chunk - 1) / pipe_chunk;
        if ((rc = ensure(ctx, ctx->out0, (nparts + 1) * 2 * L + 64)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->out1, nparts + 32)) != ECGPU_OK) return rc;
        uint8_t* part_xy = (uint8_t*)ctx->out0.p + (2 * L + 15) / 16 * 16;          // [0] is the final result
        uint8_t* part_inf = (uint8_t*)ctx->out1.p + 16;
        return dispatch(curve, [&](auto c) -> int {
            using C = decltype(c);
            int r = pipelined(ctx, n, {{scalars, &ctx->in0, L}, {points_xy, &ctx->in1, 2 * L}, {points_inf, &ctx->in2, 1}}, {},
                              [&](size_t off, size_t m) {
                                  const size_t j = off / pipe_chunk;
                                  return msm_dev<C>(ctx, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L,
                                                    points_inf ? (uint8_t*)ctx->in2.p + off : nullptr, m, part_xy + j * 2 * L, part_inf + j);
                              },
                              pipe_chunk);
            if (r != ECGPU_OK) return r;
            if ((r = point_sum_dev<C>(ctx, part_xy, part_inf, nparts, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return r;
            if ((r = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return r;
            return download(ctx, out_inf, ctx->out1, 1);
        });
    }
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (points_inf && (rc = upload(ctx, ctx->in2, points_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_msm_dev(ctx, curve, ctx->in0.p, ctx->in1.p, points_inf ? ctx->in2.p : nullptr, n, ctx->out0.p,
                            ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, 1);
}

int ecgpu_lincomb_ct(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf, size_t n,
                     uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (!out_xy || (n && (!scalars || !points_xy))) return arg_error(ctx, __func__);
    CtWipe wipe(ctx, WIPE_STAGING);
    int rc;
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (points_inf && (rc = upload(ctx, ctx->in2, points_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_lincomb_ct_dev(ctx, curve, ctx->in0.p, ctx->in1.p, points_inf ? ctx->in2.p : nullptr, n, ctx->out0.p,
                                   ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, 1);
}

int ecgpu_msm_compressed(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_x, const uint8_t* points_tag,
                         size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (!out_xy || (n && (!scalars || !points_x || !points_tag))) return arg_error(ctx, __func__);
    int rc;
    const size_t pipe_chunk = msm_pipe_chunk();
    if (n >= 2 * pipe_chunk) {          // as ecgpu_msm: one partial MSM per chunk under the upload of the next, then a point sum
        const size_t nparts = (n + pipe_chunk - 1) / pipe_chunk;
        if ((rc = ensure(ctx, ctx->out0, (nparts + 1) * 2 * L + 64)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->out1, nparts + 32)) != ECGPU_OK) return rc;
        uint8_t* part_xy = (uint8_t*)ctx->out0.p + (2 * L + 15) / 16 * 16;
        uint8_t* part_inf = (uint8_t*)ctx->out1.p + 16;
        return dispatch(curve, [&](auto c) -> int {
            using C = decltype(c);
            int r = pipelined(ctx, n, {{scalars, &ctx->in0, L}, {points_x, &ctx->in1, L}, {points_tag, &ctx->in2, 1}}, {},
                              [&](size_t off, size_t m) {
                                  const size_t j = off / pipe_chunk;
                                  return msm_compressed_dev<C>(ctx, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in1.p + off * L,
                                                               (uint8_t*)ctx->in2.p + off, m, part_xy + j * 2 * L, part_inf + j);
                              },
                              pipe_chunk);
            if (r != ECGPU_OK) return r;
            if ((r = point_sum_dev<C>(ctx, part_xy, part_inf, nparts, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return r;
            if ((r = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return r;
            return download(ctx, out_inf, ctx->out1, 1);
        });
    }
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_x, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, points_tag, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_msm_compressed_dev(ctx, curve, ctx->in0.p, ctx->in1.p, ctx->in2.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, 1);
}

int ecgpu_batch_mul_compressed(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_x, const uint8_t* points_tag,
                               size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!scalars || !points_x || !points_tag || !out_xy)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{scalars, &ctx->in0, L}, {points_x, &ctx->in1, L}, {points_tag, &ctx->in2, 1}},
                         {{out_xy, &ctx->out0, 2 * L}, {out_inf, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_batch_mul_compressed_dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L,
                                                                   (uint8_t*)ctx->in1.p + off * L, (uint8_t*)ctx->in2.p + off, m,
                                                                   (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_x, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, points_tag, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_batch_mul_compressed_dev(ctx, curve, ctx->in0.p, ctx->in1.p, ctx->in2.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_batch_mul_base_and_mul_add(ecgpu_ctx* ctx, int curve, const uint8_t* a_scalars, const uint8_t* b_scalars,
                                     const uint8_t* points_xy, const uint8_t* points_inf, size_t n, uint8_t* out_xy,
                                     uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!a_scalars || !b_scalars || !points_xy || !out_xy)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{a_scalars, &ctx->in0, L}, {b_scalars, &ctx->in3, L}, {points_xy, &ctx->in1, 2 * L}, {points_inf, &ctx->in2, 1}},
                         {{out_xy, &ctx->out0, 2 * L}, {out_inf, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_batch_mul_base_and_mul_add_dev(
                                 ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in3.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L,
                                 points_inf ? (uint8_t*)ctx->in2.p + off : nullptr, m, (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, a_scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, b_scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (points_inf && (rc = upload(ctx, ctx->in2, points_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_batch_mul_base_and_mul_add_dev(ctx, curve, ctx->in0.p, ctx->in3.p, ctx->in1.p,
                                                   points_inf ? ctx->in2.p : nullptr, n, ctx->out0.p, ctx->out1.p)) !=
        ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_ecdsa_verify_batch(ecgpu_ctx* ctx, int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s,
                             const uint8_t* q_xy, size_t n, int reject_high_s, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!z || !r || !s || !q_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{z, &ctx->in0, L}, {r, &ctx->in3, L}, {s, &ctx->in2, L}, {q_xy, &ctx->in1, 2 * L}}, {{ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return ecgpu_ecdsa_verify_batch_dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in3.p + off * L,
                                                                 (uint8_t*)ctx->in2.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L, m,
                                                                 reject_high_s, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, z, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, r, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, s, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_ecdsa_verify_batch_dev(ctx, curve, ctx->in0.p, ctx->in3.p, ctx->in2.p, ctx->in1.p, n, reject_high_s,
                                           ctx->out1.p)) != ECGPU_OK)
        return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_ecdsa_verify_msg_batch(ecgpu_ctx* ctx, int curve, const uint8_t* q_xy, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs,
                                 size_t n, int reject_high_s, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!q_xy || !sigs || !ok || (msg_len && !msgs))) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{q_xy, &ctx->in1, 2 * L}, {msg_len ? msgs : nullptr, &ctx->in0, msg_len}, {sigs, &ctx->in3, 2 * L}},
                         {{ok, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_ecdsa_verify_msg_batch_dev(ctx, curve, (uint8_t*)ctx->in1.p + off * 2 * L,
                                                                     msg_len ? (uint8_t*)ctx->in0.p + off * msg_len : nullptr, msg_len,
                                                                     (uint8_t*)ctx->in3.p + off * 2 * L, m, reject_high_s, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in0, msgs, n * msg_len)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, sigs, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_ecdsa_verify_msg_batch_dev(ctx, curve, ctx->in1.p, ctx->in0.p, msg_len, ctx->in3.p, n, reject_high_s, ctx->out1.p)) !=
        ECGPU_OK)
        return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_ecdsa_recover_batch(ecgpu_ctx* ctx, int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s,
                              const uint8_t* recid, size_t n, int reject_high_s, uint8_t* out_xy, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!z || !r || !s || !recid || !out_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{z, &ctx->in0, L}, {r, &ctx->in3, L}, {s, &ctx->in1, L}, {recid, &ctx->in2, 1}},
                         {{out_xy, &ctx->out0, 2 * L}, {ok, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_ecdsa_recover_batch_dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in3.p + off * L,
                                                                  (uint8_t*)ctx->in1.p + off * L, (uint8_t*)ctx->in2.p + off, m, reject_high_s,
                                                                  (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, z, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, r, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, s, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, recid, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_ecdsa_recover_batch_dev(ctx, curve, ctx->in0.p, ctx->in3.p, ctx->in1.p, ctx->in2.p, n, reject_high_s, ctx->out0.p,
                                            ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_sm2dsa_verify_batch(ecgpu_ctx* ctx, const uint8_t* e, const uint8_t* r, const uint8_t* s, const uint8_t* q_xy, size_t n,
                              uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    const size_t L = 32;
    if (n && (!e || !r || !s || !q_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{e, &ctx->in0, L}, {r, &ctx->in3, L}, {s, &ctx->in2, L}, {q_xy, &ctx->in1, 2 * L}}, {{ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return ecgpu_sm2dsa_verify_batch_dev(ctx, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in3.p + off * L,
                                                                  (uint8_t*)ctx->in2.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L, m,
                                                                  (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, e, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, r, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, s, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_sm2dsa_verify_batch_dev(ctx, ctx->in0.p, ctx->in3.p, ctx->in2.p, ctx->in1.p, n, ctx->out1.p)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_sm2dsa_verify_msg_batch(ecgpu_ctx* ctx, const uint8_t* distid, size_t distid_len, const uint8_t* q_xy, const uint8_t* msgs,
                                  size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    if (distid_len > 8191 || (distid_len && !distid) || (n && (!q_xy || !sigs || !ok || (msg_len && !msgs)))) return arg_error(ctx, __func__);
    int rc;
    if ((rc = upload(ctx, ctx->ec_id, distid, distid_len)) != ECGPU_OK) return rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{q_xy, &ctx->in1, 64}, {msg_len ? msgs : nullptr, &ctx->in0, msg_len}, {sigs, &ctx->in3, 64}},
                         {{ok, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_sm2dsa_verify_msg_batch_dev(ctx, ctx->ec_id.p, distid_len, (uint8_t*)ctx->in1.p + off * 64,
                                                                      msg_len ? (uint8_t*)ctx->in0.p + off * msg_len : nullptr, msg_len,
                                                                      (uint8_t*)ctx->in3.p + off * 64, m, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 64)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in0, msgs, n * msg_len)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, sigs, n * 64)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_sm2dsa_verify_msg_batch_dev(ctx, ctx->ec_id.p, distid_len, ctx->in1.p, ctx->in0.p, msg_len, ctx->in3.p, n, ctx->out1.p)) !=
        ECGPU_OK)
        return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_bign_verify_batch(ecgpu_ctx* ctx, const uint8_t* h, const uint8_t* sigs, const uint8_t* q_xy, size_t n, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    if (n && (!h || !sigs || !q_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{h, &ctx->in0, 32}, {sigs, &ctx->in3, 48}, {q_xy, &ctx->in1, 64}}, {{ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return ecgpu_bign_verify_batch_dev(ctx, (uint8_t*)ctx->in0.p + off * 32, (uint8_t*)ctx->in3.p + off * 48,
                                                                (uint8_t*)ctx->in1.p + off * 64, m, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, h, n * 32)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, sigs, n * 48)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 64)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_bign_verify_batch_dev(ctx, ctx->in0.p, ctx->in3.p, ctx->in1.p, n, ctx->out1.p)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_bign_verify_msg_batch(ecgpu_ctx* ctx, const uint8_t* q_xy, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs, size_t n,
                                uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    if (n && (!q_xy || !sigs || !ok || (msg_len && !msgs))) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{q_xy, &ctx->in1, 64}, {msg_len ? msgs : nullptr, &ctx->in0, msg_len}, {sigs, &ctx->in3, 48}},
                         {{ok, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_bign_verify_msg_batch_dev(ctx, (uint8_t*)ctx->in1.p + off * 64,
                                                                    msg_len ? (uint8_t*)ctx->in0.p + off * msg_len : nullptr, msg_len,
                                                                    (uint8_t*)ctx->in3.p + off * 48, m, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in1, q_xy, n * 64)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in0, msgs, n * msg_len)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, sigs, n * 48)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_bign_verify_msg_batch_dev(ctx, ctx->in1.p, ctx->in0.p, msg_len, ctx->in3.p, n, ctx->out1.p)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_schnorr_verify_batch(ecgpu_ctx* ctx, const uint8_t* e, const uint8_t* r, const uint8_t* s, const uint8_t* p_xy,
                               size_t n, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    const size_t L = 32;
    if (n && (!e || !r || !s || !p_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{e, &ctx->in0, L}, {r, &ctx->in3, L}, {s, &ctx->in2, L}, {p_xy, &ctx->in1, 2 * L}}, {{ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return ecgpu_schnorr_verify_batch_dev(ctx, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in3.p + off * L,
                                                                   (uint8_t*)ctx->in2.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L, m,
                                                                   (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, e, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, r, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, s, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, p_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_schnorr_verify_batch_dev(ctx, ctx->in0.p, ctx->in3.p, ctx->in2.p, ctx->in1.p, n, ctx->out1.p)) != ECGPU_OK)
        return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_schnorr_verify_raw_batch(ecgpu_ctx* ctx, const uint8_t* pk_x, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs,
                                   size_t n, uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    if (n && (!pk_x || !sigs || !ok || (msg_len && !msgs))) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{pk_x, &ctx->in0, 32}, {msg_len ? msgs : nullptr, &ctx->in1, msg_len}, {sigs, &ctx->in3, 64}},
                         {{ok, &ctx->out1, 1}}, [&](size_t off, size_t m) {
                             return ecgpu_schnorr_verify_raw_batch_dev(ctx, (uint8_t*)ctx->in0.p + off * 32,
                                                                       msg_len ? (uint8_t*)ctx->in1.p + off * msg_len : nullptr, msg_len,
                                                                       (uint8_t*)ctx->in3.p + off * 64, m, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, pk_x, n * 32)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, msgs, n * msg_len)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in3, sigs, n * 64)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_schnorr_verify_raw_batch_dev(ctx, ctx->in0.p, ctx->in1.p, msg_len, ctx->in3.p, n, ctx->out1.p)) != ECGPU_OK)
        return rc;
    return download(ctx, ok, ctx->out1, n);
}

static int batch_ecdh_host(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, size_t n, uint8_t* out_x,
                           uint8_t* ok, bool ct, const char* fn) {
    const auto dev = ct ? ecgpu_batch_ecdh_ct_dev : ecgpu_batch_ecdh_dev;
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, fn);
    if (n && (!scalars || !points_xy || !out_x || !ok)) return arg_error(ctx, fn);
    CtWipe wipe(ctx, ct ? WIPE_STAGING : 0);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{scalars, &ctx->in0, L}, {points_xy, &ctx->in1, 2 * L}}, {{out_x, &ctx->out0, L}, {ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in1.p + off * 2 * L, m,
                                                         (uint8_t*)ctx->out0.p + off * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, scalars, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = dev(ctx, curve, ctx->in0.p, ctx->in1.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_x, ctx->out0, n * L)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_batch_ecdh(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, size_t n, uint8_t* out_x, uint8_t* ok) {
    return batch_ecdh_host(ctx, curve, scalars, points_xy, n, out_x, ok, false, __func__);
}
int ecgpu_batch_ecdh_ct(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points_xy, size_t n, uint8_t* out_x,
                        uint8_t* ok) {
    return batch_ecdh_host(ctx, curve, scalars, points_xy, n, out_x, ok, true, __func__);
}

int ecgpu_batch_decompress(ecgpu_ctx* ctx, int curve, const uint8_t* xs, const uint8_t* y_is_odd, size_t n, uint8_t* out_xy,
                           uint8_t* ok) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!xs || !y_is_odd || !out_xy || !ok)) return arg_error(ctx, __func__);
    int rc;
    if (n >= PIPE_MIN)
        return pipelined(ctx, n, {{xs, &ctx->in0, L}, {y_is_odd, &ctx->in2, 1}}, {{out_xy, &ctx->out0, 2 * L}, {ok, &ctx->out1, 1}},
                         [&](size_t off, size_t m) {
                             return ecgpu_batch_decompress_dev(ctx, curve, (uint8_t*)ctx->in0.p + off * L, (uint8_t*)ctx->in2.p + off, m,
                                                               (uint8_t*)ctx->out0.p + off * 2 * L, (uint8_t*)ctx->out1.p + off);
                         });
    if ((rc = upload(ctx, ctx->in0, xs, n * L)) != ECGPU_OK) return rc;
    if ((rc = upload(ctx, ctx->in2, y_is_odd, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_batch_decompress_dev(ctx, curve, ctx->in0.p, ctx->in2.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, ok, ctx->out1, n);
}

int ecgpu_batch_normalize(ecgpu_ctx* ctx, int curve, const uint8_t* points_xyz, size_t n, uint8_t* out_xy,
                          uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!points_xyz || !out_xy)) return arg_error(ctx, __func__);
    int rc;
    if ((rc = upload(ctx, ctx->in0, points_xyz, n * 3 * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_batch_normalize_dev(ctx, curve, ctx->in0.p, n, ctx->out0.p, ctx->out1.p)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_point_sum(ecgpu_ctx* ctx, int curve, const uint8_t* points_xy, const uint8_t* points_inf, size_t n,
                    uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (!out_xy || (n && !points_xy)) return arg_error(ctx, __func__);
    int rc;
    if ((rc = upload(ctx, ctx->in1, points_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (points_inf && (rc = upload(ctx, ctx->in2, points_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, 16)) != ECGPU_OK) return rc;
    if ((rc = ecgpu_point_sum_dev(ctx, curve, ctx->in1.p, points_inf ? ctx->in2.p : nullptr, n, ctx->out0.p,
                                  ctx->out1.p)) != ECGPU_OK)
        return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, 1);
}

int ecgpu_k256_glv_decompose(ecgpu_ctx* ctx, const uint8_t* scalars, size_t n, uint8_t* r1, uint8_t* r2) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    if (n && (!scalars || !r1 || !r2)) return arg_error(ctx, __func__);
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = upload(ctx, ctx->in0, scalars, n * 32)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 32)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->in1, n * 32)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    launch_k256_glv(ctx->stream, (const uint8_t*)ctx->in0.p, n, (uint8_t*)ctx->out0.p, (uint8_t*)ctx->in1.p, ctx->d_status);
    if ((rc = finish(ctx)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, r1, ctx->out0, n * 32)) != ECGPU_OK) return rc;
    return download(ctx, r2, ctx->in1, n * 32);
}

int ecgpu_selftest_field(ecgpu_ctx* ctx, int curve, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!a || !out)) return arg_error(ctx, __func__);
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = upload(ctx, ctx->in0, a, n * L)) != ECGPU_OK) return rc;
    if (b && (rc = upload(ctx, ctx->in1, b, n * L)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * L + 16)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    rc = dispatch(curve, [&](auto c) {
        launch_selftest_field<decltype(c)>(ctx->stream, op, (const uint8_t*)ctx->in0.p, b ? (const uint8_t*)ctx->in1.p : nullptr, n,
                                           (uint8_t*)ctx->out0.p, ctx->d_status);
        return (int)ECGPU_OK;
    });
    if (rc != ECGPU_OK) return rc;
    if ((rc = finish(ctx)) != ECGPU_OK) return rc;
    return download(ctx, out, ctx->out0, n * L);
}

int ecgpu_selftest_point(ecgpu_ctx* ctx, int curve, int op, const uint8_t* p_xy, const uint8_t* p_inf, const uint8_t* q_xy,
                         const uint8_t* q_inf, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!check_ctx(ctx)) return ECGPU_ERR_ARG;
    SyncScope sync_scope(ctx);
    if (sync_scope.rc != ECGPU_OK) return sync_scope.rc;
    size_t L = ecgpu_field_bytes(curve);
    if (!L) return curve_error(ctx, __func__);
    if (n && (!p_xy || !out_xy || !out_inf)) return arg_error(ctx, __func__);
    if (n == 0) return ECGPU_OK;
    int rc;
    if ((rc = upload(ctx, ctx->in0, p_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (p_inf && (rc = upload(ctx, ctx->in2, p_inf, n)) != ECGPU_OK) return rc;
    if (q_xy && (rc = upload(ctx, ctx->in1, q_xy, n * 2 * L)) != ECGPU_OK) return rc;
    if (q_inf && (rc = upload(ctx, ctx->in3, q_inf, n)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out0, n * 2 * L + 16)) != ECGPU_OK) return rc;
    if ((rc = ensure(ctx, ctx->out1, n + 16)) != ECGPU_OK) return rc;
    if ((rc = reset_status(ctx)) != ECGPU_OK) return rc;
    rc = dispatch(curve, [&](auto c) {
        launch_selftest_point<decltype(c)>(ctx->stream, op, (const uint8_t*)ctx->in0.p, p_inf ? (const uint8_t*)ctx->in2.p : nullptr,
                                           q_xy ? (const uint8_t*)ctx->in1.p : nullptr, q_inf ? (const uint8_t*)ctx->in3.p : nullptr, n,
                                           (uint8_t*)ctx->out0.p, (uint8_t*)ctx->out1.p, ctx->d_status);
        return (int)ECGPU_OK;
    });
    if (rc != ECGPU_OK) return rc;
    if ((rc = finish(ctx)) != ECGPU_OK) return rc;
    if ((rc = download(ctx, out_xy, ctx->out0, n * 2 * L)) != ECGPU_OK) return rc;
    return download(ctx, out_inf, ctx->out1, n);
}

int ecgpu_valu_probe(ecgpu_ctx* ctx, int which, double* ops_per_sec) {
    if (!check_ctx(ctx) || !ops_per_sec) return ECGPU_ERR_ARG;
    int rc;
    if (which == 200) {
        // random 64-byte gathers over the k256 comb table (2^20 lanes x 16 entries): returns bytes per second
        if ((rc = ensure_table<K256Params>(ctx)) != ECGPU_OK) return rc;
        const Table& t = ctx->table[ECGPU_K256];
        const int gblocks = 4096, per_lane = 16;
        const size_t entries = ((size_t)1 << (t.w - 1)) * t.nwin;
        if ((rc = ensure(ctx, ctx->out0, (size_t)gblocks * BLOCK * 4)) != ECGPU_OK) return rc;
        launch_gather_probe(ctx->stream, (const uint32_t*)t.d, entries, 2, (uint32_t*)ctx->out0.p, gblocks);
        record(ctx, 0);
        launch_gather_probe(ctx->stream, (const uint32_t*)t.d, entries, per_lane, (uint32_t*)ctx->out0.p, gblocks);
        record(ctx, 1);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float gms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&gms, ctx->ev[0], ctx->ev[1]));
        *ops_per_sec = (double)gblocks * BLOCK * per_lane * 64.0 / (gms * 1e-3);
        return ECGPU_OK;
    }
    if (which == 201 || which == 202) {
        // the per-lane table pattern of the variable-base kernels with known useful bytes (ecgpu_misc.hip k_tabrow_probe):
        // 201 = every lane of a wave reads the same entry (contiguous 256-byte rows), 202 = a per-lane entry.  2048 workgroups
        // like a 2^20-element variable-base call, 64 entry reads of 20 rows per lane: returns useful bytes read per second.
        const int tblocks = 2048, reps = 64;
        const size_t lanes = (size_t)tblocks * BLOCK;
        if ((rc = ensure(ctx, ctx->vtab, lanes * 8 * 30 * 4)) != ECGPU_OK) return rc;
        if ((rc = ensure(ctx, ctx->out0, lanes * 4)) != ECGPU_OK) return rc;
        record(ctx, 0);
        launch_tabrow_probe(ctx->stream, (uint32_t*)ctx->vtab.p, which == 202, reps, (uint32_t*)ctx->out0.p, tblocks);
        record(ctx, 1);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float tms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&tms, ctx->ev[0], ctx->ev[1]));
        *ops_per_sec = (double)lanes * reps * 20 * 4.0 / (tms * 1e-3);
        return ECGPU_OK;
    }
    const int blocks = 256 * 8, iters = 2048;
    if ((rc = ensure(ctx, ctx->out0, (size_t)blocks * BLOCK * 4)) != ECGPU_OK) return rc;
    // which >= 100: exact inline-asm instruction probes (which - 100 selects the instruction, see ecgpu_misc.hip);
    // the result is then wave64-instructions per second x 64 (i.e. lane-operations per second).
    auto launch = [&](int it) {
        if (which >= 100) launch_isa_probe(ctx->stream, which - 100, (uint32_t*)ctx->out0.p, blocks, it);
        else launch_valu_probe(ctx->stream, which, (uint32_t*)ctx->out0.p, blocks, it);
    };
    launch(16);  // warm-up
    record(ctx, 0);
    launch(iters);
    record(ctx, 1);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    double ops = (double)blocks * BLOCK * (double)iters * 64.0;
    *ops_per_sec = ops / (ms * 1e-3);
    return ECGPU_OK;
}

}  // extern "C"
