// ecgpu_group.hip — the multi-GPU entry points of include/ecgpu.h: one context per device, one worker thread per device
// for the duration of a call, built on the single-GPU C ABI (ecgpu_api.hip) and the HIP runtime only.
//
//   batch workloads   index range cut into one contiguous slice per GPU; no exchange
//   MSM               terms cut into one contiguous shard per GPU; every GPU runs ecgpu_msm_parts_dev on its shard, ONE
//                     exchange step moves the per-window partial sums (tens of KiB) to where they are combined, and
//                     ecgpu_msm_finish_dev runs the window sums + the Horner chain once
//
// Exchange: RCCL's ncclAllGather over xGMI when librccl can be loaded (dlopen, no link-time dependency: a process that
// already carries torch's RCCL keeps exactly one copy) and the group's devices are distinct; otherwise — and with
// ECGPU_GROUP_EXCHANGE=peer — a peer copy of every GPU's parts into GPU 0's buffer (hipMemcpyPeer; direct over xGMI once
// peer access is enabled).  RCCL's reductions cannot add curve points, so "all-reduce of partial bucket sums" is an
// all-gather + the device-side combine in either mode.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
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
    hipStream_t stream = nullptr;      // exchange stream: the RCCL collective or the peer copy, then synchronised
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

struct ecgpu_group {
    std::vector<Member> m;
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

// runs f(r) on one thread per member; returns the first non-zero result
template <class F>
int for_each_member(ecgpu_group* g, F&& f) {
    const int nd = (int)g->m.size();
    std::vector<int> rc(nd, ECGPU_OK);
    std::vector<std::thread> th;
    try {
        for (int r = 1; r < nd; r++) th.emplace_back([&, r] { rc[r] = f(r); });
    } catch (...) {
        for (auto& t : th) t.join();
        return fail(g, ECGPU_ERR_HIP, "could not start the per-device threads");
    }
    rc[0] = f(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < nd; r++)           // (the workers have been joined: no concurrent writer of g->err any more)
        if (rc[r] != ECGPU_OK)
            return fail(g, rc[r], std::string("device ") + std::to_string(g->m[r].device) + ": " + ecgpu_last_error(g->m[r].ctx));
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
                               hipStreamCreateWithFlags(&g->m[r].stream, hipStreamNonBlocking) != hipSuccess))
            rc = ECGPU_ERR_HIP;
        if (rc != ECGPU_OK) {
            ecgpu_group_destroy(g);
            return rc;
        }
    }
    // direct peer copies into GPU 0 (xGMI) where the topology allows it; hipMemcpyPeer stages through the host otherwise
    for (int r = 1; r < ndev; r++) {
        int can = 0;
        if (devices[r] != devices[0] && hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can &&
            hipSetDevice(devices[r]) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(devices[0], 0);          // "already enabled" is fine
    }
    (void)hipGetLastError();
    const char* mode = getenv("ECGPU_GROUP_EXCHANGE");
    const bool want_rccl = !(mode && std::strcmp(mode, "peer") == 0);
    const bool must_rccl = mode && std::strcmp(mode, "rccl") == 0;
    if (!want_rccl) {
        g->why = "peer: ECGPU_GROUP_EXCHANGE=peer";
    } else if (!distinct) {
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
    if (must_rccl && !g->use_rccl) {                                // asked for RCCL explicitly and it is not to be had
        ecgpu_group_destroy(g);
        return ECGPU_ERR_HIP;
    }
    *out = g;
    return ECGPU_OK;
}

void ecgpu_group_destroy(ecgpu_group* g) {
    if (!g) return;
    for (auto& mb : g->m) {
        if (!mb.ctx) continue;
        (void)hipSetDevice(mb.device);
        if (mb.comm && g->rccl.comm_destroy) (void)g->rccl.comm_destroy(mb.comm);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        for (void* p : {mb.d_parts, mb.d_all, mb.d_in0, mb.d_in1, mb.d_in2})
            if (p) ecgpu_dev_free(mb.ctx, p);
        ecgpu_destroy(mb.ctx);
    }
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
        if ((rc = grow(g, g->m[r], &g->m[r].d_parts, &g->m[r].parts_cap, bytes)) != ECGPU_OK) return rc;
        if ((r == 0 || g->use_rccl) && (rc = grow(g, g->m[r], &g->m[r].d_all, &g->m[r].all_cap, bytes * nd)) != ECGPU_OK) return rc;
    }
    // local halves, one thread per GPU
    rc = for_each_member(g, [&](int r) -> int {
        Member& mb = g->m[r];
        return ecgpu_msm_parts_dev(mb.ctx, curve, d_scalars[r], d_points_xy[r], d_points_inf ? d_points_inf[r] : nullptr,
                                   n_per_device[r], plan_terms, mb.d_parts);           // returns with the parts written
    });
    if (rc != ECGPU_OK) return rc;
    // The exchange step.  RCCL first where the group has it.  A collective that fails — or does not END — is not the end of the
    // call: every member waits for its exchange stream by polling it against the group's deadline (ecgpu_group_set_exchange_timeout),
    // and stops waiting at once when another member's enqueue has failed (its own collective then has no partner and would never
    // complete).  On any failure every communicator is aborted, the exchange streams — which may still hold the dead collective —
    // are replaced by fresh ones, the parts, which are still in every GPU's d_parts, travel by the peer copies below, and the group
    // stays on peer copies from then on (ecgpu_group_exchange_reason says why).
    const int fault = g_test_exchange_fault.load();
    const auto wait_stream = [&](Member& mb, const std::atomic<bool>* give_up) -> int {       // 0 done, 1 deadline / given up, -1 HIP error
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(g->exchange_timeout_s);
        for (;;) {
            const hipError_t q = hipStreamQuery(mb.stream);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) return -1;
            if ((give_up && give_up->load()) || std::chrono::steady_clock::now() >= t_end) return 1;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    };
    if (g->use_rccl || fault) {
        int nrc_seen = 0;
        bool timed_out = false;
        std::mutex nrc_mu;
        std::atomic<bool> enqueue_failed{false};
        if (fault && !g->stall_release) {
            if (hipHostMalloc(reinterpret_cast<void**>(&g->stall_release), sizeof(int), hipHostMallocMapped) != hipSuccess)
                return fail(g, ECGPU_ERR_HIP, "test hook: no page-locked flag");
        }
        if (g->stall_release) *g->stall_release = 0;
        rc = for_each_member(g, [&](int r) -> int {
            Member& mb = g->m[r];
            if (hipSetDevice(mb.device) != hipSuccess) return ECGPU_ERR_HIP;
            int nrc = 0;
            if (fault == 2 || (fault == 1 && r == 0)) {
                nrc = 1;                                              // (ncclUnhandledCudaError)
            } else if (fault) {
                int* d_flag = nullptr;
                if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d_flag), g->stall_release, 0) != hipSuccess) return ECGPU_ERR_HIP;
                hipLaunchKernelGGL(k_group_stall, dim3(1), dim3(1), 0, mb.stream, d_flag);
            } else {
                nrc = g->rccl.all_gather(mb.d_parts, mb.d_all, bytes, NCCL_UINT8, mb.comm, mb.stream);
            }
            if (nrc != 0) {
                enqueue_failed.store(true);
                std::lock_guard<std::mutex> lock(nrc_mu);
                nrc_seen = nrc;
                return ECGPU_ERR_HIP;
            }
            const int w = wait_stream(mb, &enqueue_failed);
            if (w == 1 && !enqueue_failed.load()) {
                std::lock_guard<std::mutex> lock(nrc_mu);
                timed_out = true;
            }
            return w == 0 ? ECGPU_OK : ECGPU_ERR_HIP;
        });
        if (rc != ECGPU_OK) {
            g->why = std::string("peer: ncclAllGather ") +
                     (nrc_seen ? std::string("failed (") + (g->rccl.error_string && !fault ? g->rccl.error_string(nrc_seen) : "enqueue error") + ")"
                      : timed_out ? "did not complete within " + std::to_string(g->exchange_timeout_s) + " s (communicators aborted)"
                                  : std::string("failed (HIP error on the exchange stream)"));
            g->use_rccl = false;
            g->err.clear();
            // give the communicators up (ncclCommAbort ends collectives that wait for a partner), release the test stand-ins,
            // and move to fresh streams: the old ones are destroyed now if they drained, with the group otherwise
            if (g->stall_release) *g->stall_release = 1;
            for (auto& mb : g->m) {
                (void)hipSetDevice(mb.device);
                if (mb.comm) {
                    if (g->rccl.comm_abort) (void)g->rccl.comm_abort(mb.comm);
                    else if (g->rccl.comm_destroy && hipStreamQuery(mb.stream) == hipSuccess) (void)g->rccl.comm_destroy(mb.comm);
                    mb.comm = nullptr;
                }
                hipStream_t fresh = nullptr;
                if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess)
                    return fail(g, ECGPU_ERR_HIP, "exchange fallback: no fresh stream on device " + std::to_string(mb.device));
                g->abandoned.push_back(mb.stream);
                mb.stream = fresh;
            }
            (void)hipGetLastError();
        }
    }
    if (!g->use_rccl) {
        // device-to-device copies are asynchronous with respect to the host: an explicit stream + a (polled, bounded) wait, so
        // that the combining half (on member 0's own stream) starts after every part has landed
        rc = for_each_member(g, [&](int r) -> int {
            Member& mb = g->m[r];
            if (hipSetDevice(mb.device) != hipSuccess) return ECGPU_ERR_HIP;
            uint8_t* dst = (uint8_t*)g->m[0].d_all + (size_t)r * bytes;
            hipError_t he = mb.device == g->m[0].device
                                ? hipMemcpyAsync(dst, mb.d_parts, bytes, hipMemcpyDeviceToDevice, mb.stream)
                                : hipMemcpyPeerAsync(dst, g->m[0].device, mb.d_parts, mb.device, bytes, mb.stream);
            if (he != hipSuccess) return ECGPU_ERR_HIP;
            return wait_stream(mb, nullptr) == 0 ? ECGPU_OK : ECGPU_ERR_HIP;
        });
    }
    if (rc != ECGPU_OK) return rc == ECGPU_ERR_HIP && g->err.empty() ? fail(g, rc, "exchange of the partial sums failed") : rc;
    // the combining half, once
    Member& m0 = g->m[0];
    void *d_o = nullptr, *d_f = nullptr;
    if ((rc = grow(g, m0, &m0.d_in2, &m0.in2_cap, 2 * L + 64)) != ECGPU_OK) return rc;
    d_o = m0.d_in2;
    d_f = (uint8_t*)m0.d_in2 + (2 * L + 15) / 16 * 16;
    if ((rc = ecgpu_msm_finish_dev(m0.ctx, curve, m0.d_all, nd, plan_terms, d_o, d_f)) != ECGPU_OK) return fail(g, rc, ecgpu_last_error(m0.ctx));
    if ((rc = ecgpu_copy_to_host(m0.ctx, out_xy, d_o, 2 * L)) != ECGPU_OK) return fail(g, rc, ecgpu_last_error(m0.ctx));
    if (out_inf && (rc = ecgpu_copy_to_host(m0.ctx, out_inf, d_f, 1)) != ECGPU_OK) return fail(g, rc, ecgpu_last_error(m0.ctx));
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
