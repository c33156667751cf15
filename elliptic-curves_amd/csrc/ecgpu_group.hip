// ecgpu_group.hip — the multi-GPU entry points of include/ecgpu.h: one context per device, one worker thread per device for
// the life of the group (parked on a condition variable between calls: no thread is created or joined per call), built on the
// single-GPU C ABI (ecgpu_api.hip) and the HIP runtime only.
//
//   batch workloads   index range cut into one contiguous slice per GPU; no exchange
//   MSM               terms cut into one contiguous shard per GPU; every GPU runs ecgpu_msm_parts_dev on its shard, ONE
//                     exchange step moves the per-window partial sums (tens of KiB) to where they are combined, and
//                     ecgpu_msm_finish_dev runs the window sums + the Horner chain once
//
// Exchange: RCCL's ncclAllGather over xGMI when librccl can be loaded (dlopen, no link-time dependency: a process that
// already carries torch's RCCL keeps exactly one copy) and the group's devices are distinct; otherwise — and after
// ecgpu_group_set_exchange(ECGPU_EXCHANGE_PEER) — a peer copy of every GPU's parts into GPU 0's buffer (hipMemcpyPeer; direct
// over xGMI once peer access is enabled).  RCCL's reductions cannot add curve points, so "all-reduce of partial bucket sums" is an
// all-gather + the device-side combine in either mode.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ecgpu.h"

namespace {

// the slice of RCCL's C API used here (rccl.h is not needed at build time)
typedef void* nccl_comm_t;
typedef int (*nccl_comm_init_all_fn)(nccl_comm_t*, int, const int*);
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*nccl_comm_destroy_fn)(nccl_comm_t);
typedef int (*nccl_comm_abort_fn)(nccl_comm_t);
typedef const char* (*nccl_error_string_fn)(int);
constexpr int NCCL_UINT8 = 1;      // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void* lib = nullptr;
    nccl_comm_init_all_fn comm_init_all = nullptr;
    nccl_all_gather_fn all_gather = nullptr;
    nccl_comm_destroy_fn comm_destroy = nullptr;
    nccl_comm_abort_fn comm_abort = nullptr;           // optional: how a communicator with a collective that will never end is given up
    nccl_error_string_fn error_string = nullptr;
    bool load() {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        comm_init_all = (nccl_comm_init_all_fn)dlsym(lib, "ncclCommInitAll");
        all_gather = (nccl_all_gather_fn)dlsym(lib, "ncclAllGather");
        comm_destroy = (nccl_comm_destroy_fn)dlsym(lib, "ncclCommDestroy");
        comm_abort = (nccl_comm_abort_fn)dlsym(lib, "ncclCommAbort");
        error_string = (nccl_error_string_fn)dlsym(lib, "ncclGetErrorString");
        return comm_init_all && all_gather && comm_destroy;
    }
};

struct Member {
    int device = 0;
    ecgpu_ctx* ctx = nullptr;
    hipStream_t work = nullptr;        // the context's stream for the life of the group (ecgpu_set_stream): local half, combining half
    hipStream_t stream = nullptr;      // exchange stream: the RCCL collective or the peer copy, queued when the local half has ended
    hipEvent_t ev_parts = nullptr;     // recorded on `work` behind the local half
    bool leak_parts = false;           // a collective that could not be ended may still read d_parts: that allocation is never freed
    nccl_comm_t comm = nullptr;
    void* d_parts = nullptr;           // this GPU's parts record
    size_t parts_cap = 0;
    void* d_all = nullptr;             // gathered records (every member with RCCL; member 0 only with peer copies)
    size_t all_cap = 0;
    void *d_in0 = nullptr, *d_in1 = nullptr, *d_in2 = nullptr;   // shard inputs of the host-pointer MSM
    size_t in0_cap = 0, in1_cap = 0, in2_cap = 0;
};

// test-only fault injection for the exchange step (exported as ecgpu_testhook_group_exchange, not in include/ecgpu.h; no
// environment variable): 0 off; 1 the collective's enqueue fails on member 0 only while the other members' collectives are
// enqueued and never complete (the partial failure); 2 the enqueue fails on every member; 3 the collective is enqueued
// everywhere and never completes (the failure RCCL has actually shown on this pool: a hang).  With a hook set the RCCL leg is
// entered whatever the group's exchange is, and stand-ins take the place of ncclAllGather.
std::atomic<int> g_test_exchange_fault{0};

// the stand-in for a collective that never completes: spins until the host releases it (the group does when it gives the
// exchange up, so the GPU is never left with a kernel that cannot end)
__global__ void k_group_stall(volatile int* release) {
    while (!*release) __builtin_amdgcn_s_sleep(64);
}

}  // namespace

// the worker thread of member r >= 1: parked on `cv`, runs the job it is handed with its member index, reports `rc`
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    const std::function<int(int)>* job = nullptr;
    bool done = false, quit = false;
    int rc = 0;
};

struct ecgpu_group {
    std::vector<Member> m;
    std::vector<std::unique_ptr<Worker>> workers;      // workers[r - 1] serves member r (member 0 runs on the calling thread)
    uint8_t* h_out = nullptr;          // page-locked staging of the MSM's result record (one asynchronous copy behind the last kernel)
    bool use_rccl = false;
    std::string why;                   // how the exchange was chosen (ecgpu_group_exchange_reason)
    Rccl rccl;
    double exchange_timeout_s = 10.0;  // ecgpu_group_set_exchange_timeout
    int* stall_release = nullptr;      // page-locked flag of k_group_stall (test hook)
    std::vector<hipStream_t> abandoned; // exchange streams given up with work still on them (destroyed with the group if they drained)
    std::string err;
    std::mutex err_mu;                 // fail() may be called from several per-device worker threads at once
};

namespace {

int fail(ecgpu_group* g, int rc, const std::string& msg) {
    std::lock_guard<std::mutex> lock(g->err_mu);
    if (g->err.empty()) g->err = msg;          // the first failure of a call is the one reported
    return rc;
}

int grow(ecgpu_group* g, Member& mb, void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return ECGPU_OK;
    if (*p) ecgpu_dev_free(mb.ctx, *p);
    *cap = 0;
    *p = ecgpu_dev_alloc(mb.ctx, bytes + bytes / 8 + 256);
    if (!*p) return fail(g, ECGPU_ERR_OOM, std::string("device ") + std::to_string(mb.device) + ": " + ecgpu_last_error(mb.ctx));
    *cap = bytes + bytes / 8 + 256;
    return ECGPU_OK;
}

void shard(size_t n, int r, int world, size_t* lo, size_t* hi) {      // contiguous, balanced (sharded.py shard_range)
    const size_t base = n / world, rem = n % world;
    *lo = (size_t)r * base + ((size_t)r < rem ? (size_t)r : rem);
    *hi = *lo + base + ((size_t)r < rem ? 1 : 0);
}

constexpr int RELEASED = -1000;          // internal: a member's job stopped because another member had failed

void worker_main(Worker* wp, int r, int device) {
    Worker& w = *wp;
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w.mu);
    for (;;) {
        w.cv.wait(lk, [&] { return w.job != nullptr || w.quit; });
        if (w.quit) return;
        const std::function<int(int)>* job = w.job;
        lk.unlock();
        int rc;
        try {
            rc = (*job)(r);
        } catch (...) {
            rc = ECGPU_ERR_HIP;
        }
        lk.lock();
        w.rc = rc;
        w.job = nullptr;
        w.done = true;
        w.cv.notify_all();
    }
}

// runs f(r) for every member — member 0 on the calling thread, the others on their parked worker threads — and returns when all
// of them have; the first non-zero result is the call's
template <class F>
int for_each_member(ecgpu_group* g, F&& f) {
    const int nd = (int)g->m.size();
    std::vector<int> rc(nd, ECGPU_OK);
    const std::function<int(int)> job = [&f](int r) -> int { return f(r); };
    for (int r = 1; r < nd; r++) {
        Worker& w = *g->workers[r - 1];
        std::lock_guard<std::mutex> lk(w.mu);
        w.done = false;
        w.job = &job;
        w.cv.notify_all();
    }
    rc[0] = f(0);
    for (int r = 1; r < nd; r++) {
        Worker& w = *g->workers[r - 1];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        rc[r] = w.rc;
    }
    // (every worker is parked again: no concurrent writer of g->err any more.)  A member that only stopped because another one had
    // failed (RELEASED) does not decide the call's result: the member with the real error does.
    for (int pass = 0; pass < 2; pass++)
        for (int r = 0; r < nd; r++)
            if (rc[r] != ECGPU_OK && (pass == 1 || rc[r] != RELEASED)) {
                const char* ce = ecgpu_last_error(g->m[r].ctx);
                return fail(g, rc[r] == RELEASED ? ECGPU_ERR_HIP : rc[r], std::string("device ") + std::to_string(g->m[r].device) + ": " + (ce ? ce : ""));
            }
    return ECGPU_OK;
}

}  // namespace

extern "C" {

int ecgpu_group_init(ecgpu_group** out, const int* devices, int ndev) {
    if (!out) return ECGPU_ERR_ARG;
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return ECGPU_ERR_ARG;
    ecgpu_group* g = new (std::nothrow) ecgpu_group();
    if (!g) return ECGPU_ERR_OOM;
    g->m.resize(ndev);
    bool distinct = true;
    for (int r = 0; r < ndev; r++) {
        g->m[r].device = devices[r];
        for (int q = 0; q < r; q++) distinct = distinct && devices[q] != devices[r];
        int rc = ecgpu_init(&g->m[r].ctx, devices[r]);
        if (rc == ECGPU_OK && (hipSetDevice(devices[r]) != hipSuccess ||
                               hipStreamCreateWithFlags(&g->m[r].stream, hipStreamNonBlocking) != hipSuccess ||
                               hipStreamCreateWithFlags(&g->m[r].work, hipStreamNonBlocking) != hipSuccess ||
                               hipEventCreateWithFlags(&g->m[r].ev_parts, hipEventDisableTiming) != hipSuccess))
            rc = ECGPU_ERR_HIP;
        // the member's context works on a stream the group knows, so that the exchange can be ordered behind the local half on the
        // device (an event) instead of by a host wait
        if (rc == ECGPU_OK) rc = ecgpu_set_stream(g->m[r].ctx, g->m[r].work);
        if (rc != ECGPU_OK) {
            ecgpu_group_destroy(g);
            return rc;
        }
    }
    try {
        g->workers.reserve(ndev);                       // (no reallocation once a worker runs; a worker only ever sees its own Worker)
        for (int r = 1; r < ndev; r++) {
            g->workers.emplace_back(new Worker());
            Worker* w = g->workers.back().get();
            w->th = std::thread(worker_main, w, r, devices[r]);
        }
    } catch (...) {
        ecgpu_group_destroy(g);
        return ECGPU_ERR_HIP;
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&g->h_out), 512, hipHostMallocDefault) != hipSuccess) {
        ecgpu_group_destroy(g);
        return ECGPU_ERR_HIP;
    }
    // direct peer copies into GPU 0 (xGMI) where the topology allows it; hipMemcpyPeer stages through the host otherwise
    for (int r = 1; r < ndev; r++) {
        int can = 0;
        if (devices[r] != devices[0] && hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can &&
            hipSetDevice(devices[r]) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(devices[0], 0);          // "already enabled" is fine
    }
    (void)hipGetLastError();
    // (no environment variable decides this: ecgpu_group_set_exchange is how a caller asks for one exchange or the other)
    if (!distinct) {
        g->why = "peer: duplicate devices in the group (RCCL wants one communicator rank per device)";
    } else if (!g->rccl.load()) {
        const char* de = dlerror();
        g->why = std::string("peer: librccl could not be loaded (") + (de ? de : "no such library") + ")";
    } else {
        std::vector<nccl_comm_t> comms(ndev, nullptr);
        const int nrc = g->rccl.comm_init_all(comms.data(), ndev, devices);
        if (nrc == 0) {
            g->use_rccl = true;
            for (int r = 0; r < ndev; r++) g->m[r].comm = comms[r];
            g->why = "rccl: ncclCommInitAll over " + std::to_string(ndev) + (ndev == 1 ? " device" : " devices");
        } else {
            g->why = std::string("peer: ncclCommInitAll failed (") +
                     (g->rccl.error_string ? g->rccl.error_string(nrc) : ("code " + std::to_string(nrc)).c_str()) + ")";
            (void)hipGetLastError();
        }
    }
    *out = g;
    return ECGPU_OK;
}

void ecgpu_group_destroy(ecgpu_group* g) {
    if (!g) return;
    for (auto& w : g->workers) {
        if (!w->th.joinable()) continue;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
            w->cv.notify_all();
        }
        w->th.join();
    }
    if (g->stall_release) *g->stall_release = 1;                   // (test stand-ins still spinning would keep their streams busy for ever)
    for (auto& mb : g->m) {
        if (!mb.ctx) continue;
        (void)hipSetDevice(mb.device);
        if (mb.comm && g->rccl.comm_destroy) (void)g->rccl.comm_destroy(mb.comm);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        if (mb.d_parts && !mb.leak_parts) ecgpu_dev_free(mb.ctx, mb.d_parts);
        for (void* p : {mb.d_all, mb.d_in0, mb.d_in1, mb.d_in2})
            if (p) ecgpu_dev_free(mb.ctx, p);
        (void)ecgpu_set_stream(mb.ctx, nullptr);               // back on its own stream before `work` goes
        ecgpu_destroy(mb.ctx);
        if (mb.work) (void)hipStreamDestroy(mb.work);
        if (mb.ev_parts) (void)hipEventDestroy(mb.ev_parts);
    }
    if (g->h_out) (void)hipHostFree(g->h_out);
    if (g->stall_release) *g->stall_release = 1;
    for (hipStream_t st : g->abandoned)
        if (hipStreamQuery(st) == hipSuccess) (void)hipStreamDestroy(st);      // (one that still holds a dead collective is left to the process)
    if (g->stall_release) (void)hipHostFree(g->stall_release);
    delete g;
}

int ecgpu_group_size(const ecgpu_group* g) { return g ? (int)g->m.size() : 0; }

ecgpu_ctx* ecgpu_group_ctx(ecgpu_group* g, int i) { return g && i >= 0 && i < (int)g->m.size() ? g->m[i].ctx : nullptr; }

const char* ecgpu_group_last_error(const ecgpu_group* g) { return g ? g->err.c_str() : "null group"; }

const char* ecgpu_group_exchange(const ecgpu_group* g) { return g && g->use_rccl ? "rccl" : "peer"; }

const char* ecgpu_group_exchange_reason(const ecgpu_group* g) { return g ? g->why.c_str() : "null group"; }

int ecgpu_group_set_exchange(ecgpu_group* g, int mode) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    if (mode == ECGPU_EXCHANGE_RCCL)          // "RCCL or nothing": an error, not a fallback, when the group is on peer copies
        return g->use_rccl ? ECGPU_OK : fail(g, ECGPU_ERR_HIP, "the group has no RCCL exchange (" + g->why + ")");
    if (mode != ECGPU_EXCHANGE_PEER) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_set_exchange: mode must be ECGPU_EXCHANGE_PEER or ECGPU_EXCHANGE_RCCL");
    if (!g->use_rccl) return ECGPU_OK;
    for (auto& mb : g->m) {                   // no collective is in flight between calls: the communicators can simply go
        (void)hipSetDevice(mb.device);
        if (mb.comm && g->rccl.comm_destroy) (void)g->rccl.comm_destroy(mb.comm);
        mb.comm = nullptr;
    }
    (void)hipGetLastError();
    g->use_rccl = false;
    g->why = "peer: ecgpu_group_set_exchange(ECGPU_EXCHANGE_PEER)";
    return ECGPU_OK;
}

int ecgpu_group_set_exchange_timeout(ecgpu_group* g, double seconds) {
    if (!g || !(seconds > 0)) return ECGPU_ERR_ARG;
    g->exchange_timeout_s = seconds;
    return ECGPU_OK;
}

// test-only (not in include/ecgpu.h): see g_test_exchange_fault
void ecgpu_testhook_group_exchange(int mode) { g_test_exchange_fault.store(mode); }

int ecgpu_group_set_msm_window(ecgpu_group* g, int window_bits) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    for (auto& mb : g->m) {
        int rc = ecgpu_set_msm_window(mb.ctx, window_bits);
        if (rc != ECGPU_OK) return fail(g, rc, ecgpu_last_error(mb.ctx));
    }
    return ECGPU_OK;
}

int ecgpu_group_msm_dev(ecgpu_group* g, int curve, const void* const* d_scalars, const void* const* d_points_xy,
                        const void* const* d_points_inf, const size_t* n_per_device, uint8_t* out_xy, uint8_t* out_inf) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (!d_scalars || !d_points_xy || !n_per_device || !out_xy) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_msm_dev: NULL argument");
    const int nd = (int)g->m.size();
    size_t plan_terms = 1;
    for (int r = 0; r < nd; r++)
        if (n_per_device[r] > plan_terms) plan_terms = n_per_device[r];
    const size_t bytes = ecgpu_msm_parts_bytes(g->m[0].ctx, curve, plan_terms);
    if (!bytes) return fail(g, ECGPU_ERR_CURVE, ecgpu_last_error(g->m[0].ctx));
    // Every member plans for itself (its own context's window override).  The parts records only add up when all of them
    // cut the scalars into the same windows and write records of the same size: a caller who reached one member through
    // ecgpu_group_ctx and changed its window is refused here instead of overflowing d_parts / mis-striding the gather.
    const int c0 = ecgpu_msm_plan_window(g->m[0].ctx, curve, plan_terms);
    for (int r = 1; r < nd; r++)
        if (ecgpu_msm_plan_window(g->m[r].ctx, curve, plan_terms) != c0 || ecgpu_msm_parts_bytes(g->m[r].ctx, curve, plan_terms) != bytes)
            return fail(g, ECGPU_ERR_ARG, "the group's members plan different MSM windows (ecgpu_set_msm_window on one member's context?): "
                                          "use ecgpu_group_set_msm_window");
    int rc;
    for (int r = 0; r < nd; r++) {
        if (g->m[r].leak_parts && bytes > g->m[r].parts_cap) {       // (growing frees the old allocation: this one stays, see below)
            g->m[r].d_parts = nullptr;
            g->m[r].parts_cap = 0;
            g->m[r].leak_parts = false;
        }
        if ((rc = grow(g, g->m[r], &g->m[r].d_parts, &g->m[r].parts_cap, bytes)) != ECGPU_OK) return rc;
        if ((r == 0 || g->use_rccl) && (rc = grow(g, g->m[r], &g->m[r].d_all, &g->m[r].all_cap, bytes * nd)) != ECGPU_OK) return rc;
    }
    // The result record of the combining half: x || y, then the flag at the next 16-byte boundary (scratch on member 0).
    Member& m0 = g->m[0];
    const size_t flag_off = (2 * L + 15) / 16 * 16, rec_bytes = flag_off + 16;
    if ((rc = grow(g, m0, &m0.d_in2, &m0.in2_cap, 2 * L + 64)) != ECGPU_OK) return rc;
    void* d_o = m0.d_in2;
    void* d_f = (uint8_t*)m0.d_in2 + flag_off;

    // One pass over the members, each on its own (parked) thread:
    //   local half (ecgpu_msm_parts_dev, queued on the member's work stream; the context is asynchronous for that long), waited for
    //   with hipEventSynchronize — compute, it ends — and its input errors collected;
    //   THEN the collective / the peer copy is queued on the exchange stream, and the host waits for that stream against the group's
    //   deadline: spinning on hipStreamQuery for the first 200 us (a 41 KiB exchange takes tens of microseconds), then in 50 us naps.
    // The exchange is queued only when nothing else of this call is outstanding on the device, and on a stream of its own: one that
    // never ENDS — how RCCL has failed on this pool — must not be able to hold anything the call still waits for.  (Round 6 first
    // queued it BEHIND the local half with an event, to save the host round trip: HIP streams share a handful of hardware queues, and
    // a collective that waits for its partners then sat in front of another member's local half in the same queue — with the test
    // hook's stand-in, for ever.)  Every member stops waiting at the deadline, or at once when another member's enqueue has failed
    // (its own collective then has no partner); the communicators are then aborted, the exchange streams replaced by fresh ones,
    // the parts — still in every GPU's d_parts — travel by peer copies, and the group stays on peer copies
    // (ecgpu_group_exchange_reason says why).
    const int fault = g_test_exchange_fault.load();
    struct Shared {
        std::atomic<bool> enqueue_failed{false};
        std::atomic<int> parts_ended{0};       // members whose local half has ended (the exchange is queued when ALL have)
        std::mutex mu;
        int nrc_seen = 0;
        bool timed_out = false;
    } sh;
    const auto wait_exchange = [&](Member& mb, const std::atomic<bool>* give_up) -> int {     // 0 done, 1 deadline / given up, -1 HIP error
        const auto t0 = std::chrono::steady_clock::now();
        const auto t_end = t0 + std::chrono::duration<double>(g->exchange_timeout_s);
        const auto t_spin = t0 + std::chrono::microseconds(200);
        for (;;) {
            const hipError_t q = hipStreamQuery(mb.stream);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) return -1;
            const auto now = std::chrono::steady_clock::now();
            if ((give_up && give_up->load()) || now >= t_end) return 1;
            if (now >= t_spin) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    };
    // with_parts: the local half is part of the pass (the first pass); use_coll: the exchange is the collective (or its stand-ins)
    const auto pass = [&](bool with_parts, bool use_coll) -> int {
        return for_each_member(g, [&](int r) -> int {
            Member& mb = g->m[r];
            const auto bail = [&](int code) { sh.enqueue_failed.store(true); return code; };   // the others must not wait for this member
            if (hipSetDevice(mb.device) != hipSuccess) return bail(ECGPU_ERR_HIP);
            int e;
            // the context is asynchronous from here to the end of its local half, whatever path leaves this function
            struct AsyncScope {
                ecgpu_ctx* ctx = nullptr;
                ~AsyncScope() { if (ctx) (void)ecgpu_set_async(ctx, 0); }
            } scope;
            if (with_parts) {
                if ((e = ecgpu_set_async(mb.ctx, 1)) != ECGPU_OK) return bail(e);
                scope.ctx = mb.ctx;
                e = ecgpu_msm_parts_dev(mb.ctx, curve, d_scalars[r], d_points_xy[r], d_points_inf ? d_points_inf[r] : nullptr, n_per_device[r],
                                        plan_terms, mb.d_parts);
                if (e == ECGPU_OK && hipEventRecord(mb.ev_parts, mb.work) != hipSuccess) e = ECGPU_ERR_HIP;
                // the local half: compute, it ends; its input errors (a scalar >= n, a point off the curve) are the call's result
                if (e == ECGPU_OK) e = hipEventSynchronize(mb.ev_parts) == hipSuccess ? ecgpu_synchronize(mb.ctx) : ECGPU_ERR_HIP;
                scope.ctx = nullptr;
                const int e2 = ecgpu_set_async(mb.ctx, 0);
                if (e == ECGPU_OK) e = e2;
                if (e != ECGPU_OK) return bail(e);
                // Nobody queues its exchange before EVERY member's local half has ended: a collective that waits for its partners in
                // a hardware queue another member's local half is still queued in (members on one device share queues) would wait
                // for ever.  A member that failed releases the others (enqueue_failed).
                sh.parts_ended.fetch_add(1);
                while (sh.parts_ended.load() < nd && !sh.enqueue_failed.load()) std::this_thread::yield();
                if (sh.enqueue_failed.load()) return RELEASED;
            }
            bool coll_failed = false;
            if (use_coll) {
                int nrc = 0;
                if (fault == 2 || (fault == 1 && r == 0)) {
                    nrc = 1;                                              // (ncclUnhandledCudaError)
                } else if (fault) {
                    int* d_flag = nullptr;
                    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d_flag), g->stall_release, 0) != hipSuccess) return bail(ECGPU_ERR_HIP);
                    hipLaunchKernelGGL(k_group_stall, dim3(1), dim3(1), 0, mb.stream, d_flag);
                } else {
                    nrc = g->rccl.all_gather(mb.d_parts, mb.d_all, bytes, NCCL_UINT8, mb.comm, mb.stream);
                }
                if (nrc != 0) {                                          // the others stop waiting at once
                    std::lock_guard<std::mutex> lock(sh.mu);
                    sh.nrc_seen = nrc;
                    sh.enqueue_failed.store(true);
                    coll_failed = true;
                }
            } else {
                uint8_t* dst = (uint8_t*)m0.d_all + (size_t)r * bytes;
                const hipError_t he = mb.device == m0.device
                                          ? hipMemcpyAsync(dst, mb.d_parts, bytes, hipMemcpyDeviceToDevice, mb.stream)
                                          : hipMemcpyPeerAsync(dst, m0.device, mb.d_parts, mb.device, bytes, mb.stream);
                if (he != hipSuccess) return bail(ECGPU_ERR_HIP);
            }
            if (coll_failed) return ECGPU_ERR_HIP;
            const int w = wait_exchange(mb, use_coll ? &sh.enqueue_failed : nullptr);
            if (w == 1 && !sh.enqueue_failed.load()) {
                std::lock_guard<std::mutex> lock(sh.mu);
                sh.timed_out = true;
            }
            return w == 0 ? ECGPU_OK : ECGPU_ERR_HIP;
        });
    };
    const bool coll = g->use_rccl || fault;
    if (fault && !g->stall_release) {
        if (hipHostMalloc(reinterpret_cast<void**>(&g->stall_release), sizeof(int), hipHostMallocMapped) != hipSuccess)
            return fail(g, ECGPU_ERR_HIP, "test hook: no page-locked flag");
    }
    if (g->stall_release) *g->stall_release = 0;
    rc = pass(true, coll);
    // An input error or a HIP failure of a local half is the call's result; only a failed EXCHANGE is retried over peer copies — when
    // every local half has ended (a member whose enqueue failed left before waiting for its own: compute, it ends).
    bool parts_done = true;
    if (rc != ECGPU_OK)
        for (auto& mb : g->m)
            parts_done = hipSetDevice(mb.device) == hipSuccess && hipEventSynchronize(mb.ev_parts) == hipSuccess && parts_done;
    if (rc != ECGPU_OK && g->stall_release) *g->stall_release = 1;     // (the stand-ins of the test hook never outlive a failed pass)
    if (rc != ECGPU_OK && coll && parts_done && (sh.nrc_seen || sh.timed_out) && rc == ECGPU_ERR_HIP) {
        const std::string why = std::string("peer: ncclAllGather ") +
                                (sh.nrc_seen ? std::string("failed (") + (g->rccl.error_string && !fault ? g->rccl.error_string(sh.nrc_seen) : "enqueue error") + ")"
                                 : sh.timed_out ? "did not complete within " + std::to_string(g->exchange_timeout_s) + " s (communicators aborted)"
                                                : std::string("failed (HIP error on the exchange stream)"));
        // fresh exchange streams for everybody BEFORE anything is given up: a failure here leaves the group as it was
        std::vector<hipStream_t> fresh(nd, nullptr);
        for (int r = 0; r < nd; r++)
            if (hipSetDevice(g->m[r].device) != hipSuccess || hipStreamCreateWithFlags(&fresh[r], hipStreamNonBlocking) != hipSuccess) {
                for (int q = 0; q < r; q++) {
                    (void)hipSetDevice(g->m[q].device);
                    (void)hipStreamDestroy(fresh[q]);
                }
                if (g->stall_release) *g->stall_release = 1;
                return fail(g, ECGPU_ERR_HIP, "exchange fallback: no fresh stream on device " + std::to_string(g->m[r].device));
            }
        g->why = why;
        g->use_rccl = false;
        {
            std::lock_guard<std::mutex> lock(g->err_mu);
            g->err.clear();
        }
        // give the communicators up (ncclCommAbort ends collectives that wait for a partner), release the test stand-ins, and move
        // to the fresh streams: the old ones are destroyed with the group if they drained
        if (g->stall_release) *g->stall_release = 1;
        for (int r = 0; r < nd; r++) {
            Member& mb = g->m[r];
            (void)hipSetDevice(mb.device);
            if (mb.comm) {
                if (g->rccl.comm_abort) {
                    (void)g->rccl.comm_abort(mb.comm);
                } else if (g->rccl.comm_destroy && hipStreamQuery(mb.stream) == hipSuccess) {
                    (void)g->rccl.comm_destroy(mb.comm);
                } else {
                    // No way to end the collective: it may still read d_parts and write d_all whenever its partners show up.  The
                    // communicator is left to the process; d_all is given up (never freed; the retry below gets a fresh destination
                    // on member 0) and d_parts — which the retry reads, as the collective does — is never freed either.
                    mb.d_all = nullptr;
                    mb.all_cap = 0;
                    mb.leak_parts = true;
                    if (r == 0 && (rc = grow(g, mb, &mb.d_all, &mb.all_cap, bytes * nd)) != ECGPU_OK) return rc;
                }
                mb.comm = nullptr;
            }
            g->abandoned.push_back(mb.stream);
            mb.stream = fresh[r];
        }
        (void)hipGetLastError();
        sh.enqueue_failed.store(false);
        sh.nrc_seen = 0;
        sh.timed_out = false;
        rc = pass(false, false);
    }
    if (rc != ECGPU_OK) {
        std::lock_guard<std::mutex> lock(g->err_mu);
        if (rc == ECGPU_ERR_HIP && g->err.empty()) g->err = "exchange of the partial sums failed";
        return rc;
    }
    // the combining half, once: queued on member 0's work stream with the copy of its record behind it — one more wait
    if (hipSetDevice(m0.device) != hipSuccess) return fail(g, ECGPU_ERR_HIP, "hipSetDevice");
    if ((rc = ecgpu_set_async(m0.ctx, 1)) != ECGPU_OK) return fail(g, rc, ecgpu_last_error(m0.ctx));
    rc = ecgpu_msm_finish_dev(m0.ctx, curve, m0.d_all, nd, plan_terms, d_o, d_f);
    if (rc == ECGPU_OK && hipMemcpyAsync(g->h_out, d_o, rec_bytes, hipMemcpyDeviceToHost, m0.work) != hipSuccess) rc = ECGPU_ERR_HIP;
    const int rs = ecgpu_synchronize(m0.ctx);
    const int ra = ecgpu_set_async(m0.ctx, 0);
    if (rc == ECGPU_OK) rc = rs != ECGPU_OK ? rs : ra;
    if (rc != ECGPU_OK) return fail(g, rc, ecgpu_last_error(m0.ctx));
    std::memcpy(out_xy, g->h_out, 2 * L);
    if (out_inf) *out_inf = g->h_out[flag_off];
    return ECGPU_OK;
}

int ecgpu_group_msm(ecgpu_group* g, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf,
                    size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (!out_xy || (n && (!scalars || !points_xy))) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_msm: NULL argument");
    const int nd = (int)g->m.size();
    std::vector<const void*> ds(nd), dp(nd), di(nd);
    std::vector<size_t> cnt(nd);
    // shard upload: every GPU pulls its own slice over its own PCIe link, in parallel
    int rc = for_each_member(g, [&](int r) -> int {
        Member& mb = g->m[r];
        size_t lo, hi;
        shard(n, r, nd, &lo, &hi);
        const size_t m = hi - lo;
        cnt[r] = m;
        int e;
        if ((e = grow(g, mb, &mb.d_in0, &mb.in0_cap, m * L + 16)) != ECGPU_OK) return e;
        if ((e = grow(g, mb, &mb.d_in1, &mb.in1_cap, m * 2 * L + 16)) != ECGPU_OK) return e;
        if ((e = ecgpu_copy_to_device(mb.ctx, mb.d_in0, scalars + lo * L, m * L)) != ECGPU_OK) return e;
        if ((e = ecgpu_copy_to_device(mb.ctx, mb.d_in1, points_xy + lo * 2 * L, m * 2 * L)) != ECGPU_OK) return e;
        ds[r] = mb.d_in0;
        dp[r] = mb.d_in1;
        di[r] = nullptr;
        if (points_inf) {
            // flags share the scratch buffer the result record later uses on member 0: keep them apart
            if ((e = grow(g, mb, &mb.d_in2, &mb.in2_cap, m + 2 * L + 128)) != ECGPU_OK) return e;
            uint8_t* flags = (uint8_t*)mb.d_in2 + (2 * L + 64 + 15) / 16 * 16;
            if ((e = ecgpu_copy_to_device(mb.ctx, flags, points_inf + lo, m)) != ECGPU_OK) return e;
            di[r] = flags;
        }
        return ECGPU_OK;
    });
    if (rc != ECGPU_OK) return rc;
    return ecgpu_group_msm_dev(g, curve, ds.data(), dp.data(), points_inf ? di.data() : nullptr, cnt.data(), out_xy, out_inf);
}

int ecgpu_group_batch_mul_base(ecgpu_group* g, int curve, const uint8_t* scalars, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (n && (!scalars || !out_xy)) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_batch_mul_base: NULL argument");
    const int nd = (int)g->m.size();
    return for_each_member(g, [&](int r) -> int {
        size_t lo, hi;
        shard(n, r, nd, &lo, &hi);
        return ecgpu_batch_mul_base(g->m[r].ctx, curve, scalars + lo * L, hi - lo, out_xy + lo * 2 * L, out_inf ? out_inf + lo : nullptr);
    });
}

int ecgpu_group_batch_mul(ecgpu_group* g, int curve, const uint8_t* scalars, const uint8_t* points_xy, const uint8_t* points_inf,
                          size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (n && (!scalars || !points_xy || !out_xy)) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_batch_mul: NULL argument");
    const int nd = (int)g->m.size();
    return for_each_member(g, [&](int r) -> int {
        size_t lo, hi;
        shard(n, r, nd, &lo, &hi);
        return ecgpu_batch_mul(g->m[r].ctx, curve, scalars + lo * L, points_xy + lo * 2 * L, points_inf ? points_inf + lo : nullptr,
                               hi - lo, out_xy + lo * 2 * L, out_inf ? out_inf + lo : nullptr);
    });
}

// ---- the signature entry points over the group: index-range slices, no exchange ----------------------------------------------
int ecgpu_group_ecdsa_verify_batch(ecgpu_group* g, int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* q_xy,
                                   size_t n, int reject_high_s, uint8_t* ok) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (n && (!z || !r || !s || !q_xy || !ok)) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_ecdsa_verify_batch: NULL argument");
    const int nd = (int)g->m.size();
    return for_each_member(g, [&](int m) -> int {
        size_t lo, hi;
        shard(n, m, nd, &lo, &hi);
        return ecgpu_ecdsa_verify_batch(g->m[m].ctx, curve, z + lo * L, r + lo * L, s + lo * L, q_xy + lo * 2 * L, hi - lo, reject_high_s,
                                        ok + lo);
    });
}

int ecgpu_group_ecdsa_verify_msg_batch(ecgpu_group* g, int curve, const uint8_t* q_xy, const uint8_t* msgs, size_t msg_len,
                                       const uint8_t* sigs, size_t n, int reject_high_s, uint8_t* ok) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (n && (!q_xy || !sigs || !ok || (msg_len && !msgs))) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_ecdsa_verify_msg_batch: NULL argument");
    const int nd = (int)g->m.size();
    return for_each_member(g, [&](int m) -> int {
        size_t lo, hi;
        shard(n, m, nd, &lo, &hi);
        return ecgpu_ecdsa_verify_msg_batch(g->m[m].ctx, curve, q_xy + lo * 2 * L, msg_len ? msgs + lo * msg_len : nullptr, msg_len,
                                            sigs + lo * 2 * L, hi - lo, reject_high_s, ok + lo);
    });
}

int ecgpu_group_ecdsa_recover_batch(ecgpu_group* g, int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* recid,
                                    size_t n, int reject_high_s, uint8_t* out_xy, uint8_t* ok) {
    if (!g) return ECGPU_ERR_ARG;
    g->err.clear();
    const size_t L = ecgpu_field_bytes(curve);
    if (!L) return fail(g, ECGPU_ERR_CURVE, "unknown curve id");
    if (n && (!z || !r || !s || !recid || !out_xy || !ok)) return fail(g, ECGPU_ERR_ARG, "ecgpu_group_ecdsa_recover_batch: NULL argument");
    const int nd = (int)g->m.size();
    return for_each_member(g, [&](int m) -> int {
        size_t lo, hi;
        shard(n, m, nd, &lo, &hi);
        return ecgpu_ecdsa_recover_batch(g->m[m].ctx, curve, z + lo * L, r + lo * L, s + lo * L, recid + lo, hi - lo, reject_high_s,
                                         out_xy + lo * 2 * L, ok + lo);
    });
}

}  // extern "C"
