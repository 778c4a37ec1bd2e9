// multi.cpp -- several MI355X behind ONE C-ABI object (rat_multi), driven from the one calling thread.
//
// Replaces the reference's process fan-out of compute_cost (cross_entropy_bilevel_optimization.jl:180-192: `@sync for ... @async
// remotecall_fetch(compute_value_worker, 2 + mod(i, nprocs - 1), ...)`), which a Julia host reaches through `addprocs`; here the host
// (Julia via ccall, Python via ctypes, C) holds one rat_multi and never sees a process group:
//   * one rat_handle per device, each with its own HIP stream (built on the public single-device entry points of include/ratilqr.h);
//   * theta-samples are split in CONTIGUOUS blocks (rat_shard_bounds; the reference deals them round-robin, :181 -- assignment does
//     not affect results), every device runs the complete solves of its block in one launch on its stream;
//   * ONE ncclAllGather (RCCL over xGMI) of the per-sample costs, ordered on those same streams inside one ncclGroup, leaves cost[B] on
//     every device -- B/G doubles per rank, latency-bound, nothing else crosses devices -- and device 0's copy goes back to the host;
//   * elite selection (rat_ce_update) is host arithmetic, the final solve at theta_opt is a single trajectory on device 0.
//   * the gathered record of a sample is its cost AND its status, iteration count and line-search count: every rank
//     contributes ONE byte block [cost f64 x chunk | status, iters, ls_evals i32 x chunk], so it is still one collective per batch.
// RCCL (librccl.so, 570 MB) is loaded with dlopen the first time a rat_multi with more than one device is created: single-GPU users of
// libratilqr_hip.so never pay for it (RATILQR_MULTI_FORCE_RCCL=1 runs the collective with a one-rank communicator: test hook).
// RATILQR_MULTI_LOGICAL=1 (test hook for a one-GPU box): the G "devices" are logical -- device index d runs on physical device d mod the
// number visible, several handles per GPU, each with its own stream -- and the all-gather is carried out by stream-ordered device copies
// into the very slots d_all[g] + r * block that RCCL would fill (RCCL refuses two ranks on one GPU).  Everything else -- shard bounds,
// ragged pads, per-device enqueue, offsets of the read-back -- is the code a real G-device node runs.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ratilqr.h"

// the handful of RCCL entry points used, resolved at run time (signatures: /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum { NCCL_SUCCESS_ = 0, NCCL_INT8_ = 0, NCCL_FLOAT64_ = 8 };      // ncclSuccess, ncclInt8 / ncclChar, ncclFloat64 / ncclDouble
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

void rat_set_error(const char *msg);                // driver.cpp: thread-local message behind rat_last_error()
bool rat_batch_is_single_launch(rat_handle h, int64_t B);   // driver.cpp
rat_rc rat_batch_outputs_dev(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, bool as_value, double *cost_dev, int32_t *status_dev,
                             int32_t *iters_dev, int32_t *ls_dev, bool wait);                    // driver.cpp
static rat_rc mfail(rat_rc rc, const std::string &m) { rat_set_error(m.c_str()); return rc; }
#define MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mfail(RAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define MRC(expr) do { rat_rc r_ = (expr); if (r_ != RAT_OK) return r_; } while (0)
// after work has been enqueued on the devices of a rat_multi `m`: a failing call waits for EVERY device before it hands control back
// (other devices' streams may still be reading d_theta / the pinned stage and writing d_all, which the caller is free to reuse or destroy)
#define MHIP_SYNC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { sync_all(m); \
        return mfail(RAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

static rat_rc load_rccl() {
    if (g_rccl.so) return RAT_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *so = nullptr;
    for (const char *n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!so) return mfail(RAT_ERR_HIP, std::string("rat_create_multi: cannot load RCCL: ") + dlerror());
    Rccl r;
    r.so = so;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(so, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(so, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(so, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(so, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return mfail(RAT_ERR_HIP, "rat_create_multi: librccl.so lacks an expected symbol");
    g_rccl = r;
    return RAT_OK;
}
#define MNCCL(expr) do { ncclResult_t e_ = (expr); if (e_ != NCCL_SUCCESS_) return mfail(RAT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(e_)); } while (0)

struct rat_multi_s {
    int G = 0;
    int Bmax = 0, chunk_max = 0;
    std::vector<int> dev;
    std::vector<rat_handle> h;
    std::vector<ncclComm_t> comm;                   // empty: no RCCL (one device)
    bool logical = false;                           // RATILQR_MULTI_LOGICAL=1: several handles per GPU, all-gather by device copies
    std::vector<hipEvent_t> ev;                     // per device: "this device's block is complete" (copy-based gather)
    std::vector<double *> d_theta;                  // per device: theta shard [chunk_max]
    std::vector<char *> d_send, d_all;              // per device: its block [block_bytes(chunk_max)], the gathered blocks [G * block_bytes(chunk_max)]
    char *h_stage = nullptr;                        // pinned: theta (Bmax doubles) | gathered blocks (G * block_bytes(chunk_max))
    double *h_pets = nullptr; size_t pets_cap = 0;  // pinned: PETS per-sample costs
    int pets_n = 0, pets_m = 0, pets_N = 0;
    int64_t n_allgathers = 0;
};

// one rank's contribution to the all-gather: [cost f64 x chunk | status i32 x chunk | iters i32 x chunk | ls_evals i32 x chunk], 8-B aligned
static inline size_t block_bytes(int64_t chunk) { return ((size_t)chunk * 20 + 7) & ~(size_t)7; }

extern "C" rat_rc rat_shard_bounds(int64_t B, int32_t world, int32_t rank, int64_t *lo, int64_t *hi) {
    if (B < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return mfail(RAT_ERR_ARG, "rat_shard_bounds: bad arguments");
    const int64_t base = B / world, rem = B % world;            // blocks differ by at most one sample; the first `rem` blocks are the longer ones
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return RAT_OK;
}

extern "C" void rat_multi_destroy(rat_multi m) {
    if (!m) return;
    for (int g = 0; g < (int)m->h.size(); ++g) {
        (void)hipSetDevice(m->dev[g]);
        if (m->h[g]) (void)hipStreamSynchronize((hipStream_t)rat_stream(m->h[g]));
        if (g < (int)m->comm.size() && m->comm[g]) (void)g_rccl.CommDestroy(m->comm[g]);
        if (g < (int)m->d_theta.size() && m->d_theta[g]) (void)hipFree(m->d_theta[g]);
        for (auto *v : {&m->d_send, &m->d_all}) if (g < (int)v->size() && (*v)[g]) (void)hipFree((*v)[g]);
        if (g < (int)m->ev.size() && m->ev[g]) (void)hipEventDestroy(m->ev[g]);
        if (m->h[g]) rat_destroy(m->h[g]);
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->h_pets) (void)hipHostFree(m->h_pets);
    delete m;
}

extern "C" rat_rc rat_create_multi(const rat_ileqg_opts *opts, int32_t max_batch, int32_t spec_eps, int32_t n_devices, const int32_t *devices,
                                   rat_multi *out) {
    if (!out || max_batch < 1 || n_devices < 1) return mfail(RAT_ERR_ARG, "rat_create_multi: bad max_batch / n_devices");
    int ndev = 0;
    MHIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return mfail(RAT_ERR_HIP, "rat_create_multi: no HIP device (this library has no CPU fallback)");
    const char *lg = getenv("RATILQR_MULTI_LOGICAL");
    const bool logical = lg && lg[0] == '1';
    if (n_devices > ndev && !logical) return mfail(RAT_ERR_ARG, "rat_create_multi: more devices requested than are visible (there is no CPU fallback)");
    rat_multi m = new rat_multi_s();
    m->logical = logical;
    m->G = n_devices; m->Bmax = max_batch; m->chunk_max = (max_batch + n_devices - 1) / n_devices;
    m->h.assign(n_devices, nullptr);
    for (int g = 0; g < n_devices; ++g) {
        int d = devices ? devices[g] : g;
        if (d < 0 || (d >= ndev && !logical)) { rat_multi_destroy(m); return mfail(RAT_ERR_ARG, "rat_create_multi: bad device index"); }
        if (!logical) for (int q = 0; q < g; ++q) if (m->dev[q] == d) { rat_multi_destroy(m); return mfail(RAT_ERR_ARG, "rat_create_multi: device listed twice"); }
        m->dev.push_back(logical ? d % ndev : d);
    }
    m->d_theta.assign(n_devices, nullptr); m->d_send.assign(n_devices, nullptr); m->d_all.assign(n_devices, nullptr);
    m->ev.assign(n_devices, nullptr);
#define MCREATE(expr) do { if ((expr) != 0) { rat_multi_destroy(m); return mfail(RAT_ERR_HIP, std::string("rat_create_multi: ") + #expr + " failed: " + rat_last_error()); } } while (0)
    for (int g = 0; g < n_devices; ++g) {
        MCREATE(rat_create(opts, m->chunk_max, spec_eps, m->dev[g], &m->h[g]));
        MCREATE((int)hipSetDevice(m->dev[g]));
        MCREATE((int)hipMalloc((void **)&m->d_theta[g], sizeof(double) * m->chunk_max));
        MCREATE((int)hipMalloc((void **)&m->d_send[g], block_bytes(m->chunk_max)));
        MCREATE((int)hipMalloc((void **)&m->d_all[g], block_bytes(m->chunk_max) * n_devices));
        MCREATE((int)hipEventCreateWithFlags(&m->ev[g], hipEventDisableTiming));
    }
    MCREATE((int)hipHostMalloc((void **)&m->h_stage, sizeof(double) * (size_t)max_batch + block_bytes(m->chunk_max) * n_devices, hipHostMallocDefault));
    const char *force = getenv("RATILQR_MULTI_FORCE_RCCL");
    if (!logical && (n_devices > 1 || (force && force[0] == '1'))) {
        rat_rc rc = load_rccl();
        if (rc) { rat_multi_destroy(m); return rc; }
        m->comm.assign(n_devices, nullptr);
        ncclResult_t e = g_rccl.CommInitAll(m->comm.data(), n_devices, m->dev.data());      // one thread, all devices
        if (e != NCCL_SUCCESS_) { m->comm.clear(); rat_multi_destroy(m); return mfail(RAT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e)); }
    }
#undef MCREATE
    *out = m;
    return RAT_OK;
}

extern "C" int32_t rat_multi_n_devices(rat_multi m) { return m ? m->G : 0; }
extern "C" rat_handle rat_multi_handle(rat_multi m, int32_t i) { return (m && i >= 0 && i < m->G) ? m->h[i] : nullptr; }
extern "C" int64_t rat_multi_allgathers(rat_multi m) { return m ? m->n_allgathers : -1; }
extern "C" int32_t rat_multi_uses_rccl(rat_multi m) { return (m && !m->comm.empty()) ? 1 : 0; }
extern "C" int32_t rat_multi_is_logical(rat_multi m) { return (m && m->logical) ? 1 : 0; }

extern "C" rat_rc rat_multi_problem_set(rat_multi m, const rat_problem_desc *d) {
    if (!m || !d) return mfail(RAT_ERR_ARG, "null");
    for (int g = 0; g < m->G; ++g) MRC(rat_problem_set(m->h[g], d));
    return RAT_OK;
}
extern "C" rat_rc rat_multi_set_initial(rat_multi m, const double *x0, const double *u0) {
    if (!m) return mfail(RAT_ERR_ARG, "null");
    for (int g = 0; g < m->G; ++g) MRC(rat_set_initial(m->h[g], x0, u0));
    return RAT_OK;
}

// On any failure after work has been enqueued: wait for every device before handing control back (other devices' launches are still
// reading d_theta and the pinned stage).
static void sync_all(rat_multi m) {
    for (int g = 0; g < m->G; ++g) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) continue;
        (void)hipStreamSynchronize((hipStream_t)rat_stream(m->h[g]));
    }
}

// compute_cost (cross_entropy_bilevel_optimization.jl:173-195) over all devices: cost_i = value_i + kl_bound / theta_i, +Inf for failures;
// with the per-sample status / iteration count / line-search count of every shard gathered beside the costs (any of them may be NULL).
static rat_rc multi_batch(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B, double kl_bound, bool as_value,
                          double *cost, int32_t *status, int32_t *iters, int32_t *ls_evals) {
    if (!m || !theta || !cost) return mfail(RAT_ERR_ARG, "null");
    if (B < 1 || B > m->Bmax) return mfail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create_multi");
    if (x0 && u0) MRC(rat_multi_set_initial(m, x0, u0));
    const int G = m->G;
    const int64_t chunk = (B + G - 1) / G;
    const size_t bb = block_bytes(chunk);                        // one rank's block in this batch
    double *p_theta = reinterpret_cast<double *>(m->h_stage);
    char *p_all = m->h_stage + sizeof(double) * (size_t)m->Bmax;
    memcpy(p_theta, theta, sizeof(double) * B);
    std::vector<int64_t> lo((size_t)G), hi((size_t)G);
    for (int g = 0; g < G; ++g) MRC(rat_shard_bounds(B, G, g, &lo[(size_t)g], &hi[(size_t)g]));
    auto sect = [&](int g, int which) {                          // sections of device g's block: 0 cost, 1 status, 2 iters, 3 ls_evals
        return m->d_send[g] + (which == 0 ? 0 : (size_t)chunk * 8 + (size_t)(which - 1) * (size_t)chunk * 4);
    };
    std::vector<int> polled;
    rat_rc rc = RAT_OK;
    // every device: its theta block in, its solves enqueued -- all asynchronous on the device's own stream, no host wait in this loop
    for (int g = 0; g < G && rc == RAT_OK; ++g) {
        const int64_t nb = hi[(size_t)g] - lo[(size_t)g];
        if (hipSetDevice(m->dev[g]) != hipSuccess) { rc = mfail(RAT_ERR_HIP, "hipSetDevice failed"); break; }
        hipStream_t s = (hipStream_t)rat_stream(m->h[g]);
        // pad slots of a short (or empty) block: all-ones bytes (NaN costs, -1 counters), never read back; filled on the device, in
        // stream order before the solves -- no host buffer, no wait (ragged batches only)
        if (nb < chunk && hipMemsetAsync(m->d_send[g], 0xFF, bb, s) != hipSuccess) { rc = mfail(RAT_ERR_HIP, "hipMemsetAsync failed"); break; }
        if (nb > 0) {
            if (hipMemcpyAsync(m->d_theta[g], p_theta + lo[(size_t)g], sizeof(double) * nb, hipMemcpyHostToDevice, s) != hipSuccess) {
                rc = mfail(RAT_ERR_HIP, "hipMemcpyAsync(theta) failed"); break;
            }
            if (rat_batch_is_single_launch(m->h[g], nb))
                rc = rat_batch_outputs_dev(m->h[g], m->d_theta[g], nb, kl_bound, as_value, (double *)sect(g, 0), (int32_t *)sect(g, 1), (int32_t *)sect(g, 2),
                                           (int32_t *)sect(g, 3), false);
            else polled.push_back(g);
        }
    }
    if (rc != RAT_OK) { sync_all(m); return rc; }
    if (!polled.empty()) {
        // Shards on the round-based path (speculative step sizes on a shard too large for one generation of workgroups): its host loop
        // polls the device between rounds, so one calling thread would process the devices one after the other.  A helper thread per
        // such device instead, ALWAYS joined before anything else happens; their error messages come back through rcs / msgs.
        std::vector<rat_rc> rcs(polled.size(), RAT_OK);
        std::vector<std::string> msgs(polled.size());
        {
            std::vector<std::thread> th;
            struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{th};
            th.reserve(polled.size());                           // (no reallocation, and a throwing emplace_back still joins what runs)
            for (size_t q = 0; q < polled.size(); ++q) {
                const int g = polled[q];
                const int64_t nb = hi[(size_t)g] - lo[(size_t)g];
                double *c_ = (double *)sect(g, 0); int32_t *s_ = (int32_t *)sect(g, 1), *i_ = (int32_t *)sect(g, 2), *l_ = (int32_t *)sect(g, 3);
                th.emplace_back([m, g, q, nb, kl_bound, as_value, c_, s_, i_, l_, &rcs, &msgs]() {
                    if (hipSetDevice(m->dev[g]) != hipSuccess) { rcs[q] = RAT_ERR_HIP; msgs[q] = "hipSetDevice failed in a shard thread"; return; }
                    rcs[q] = rat_batch_outputs_dev(m->h[g], m->d_theta[g], nb, kl_bound, as_value, c_, s_, i_, l_, true);
                    if (rcs[q] != RAT_OK) msgs[q] = rat_last_error();
                });
            }
        }
        for (size_t q = 0; q < polled.size(); ++q) if (rcs[q] != RAT_OK) { sync_all(m); return mfail(rcs[q], msgs[q]); }
    }
    if (!m->comm.empty()) {
        // ONE collective per batch: every rank contributes its block of `bb` bytes and receives G of them, ordered behind its solves on its stream
        ncclResult_t e = g_rccl.GroupStart();
        for (int g = 0; g < G && e == NCCL_SUCCESS_; ++g)
            e = g_rccl.AllGather(m->d_send[g], m->d_all[g], bb, NCCL_INT8_, m->comm[g], (hipStream_t)rat_stream(m->h[g]));
        const ncclResult_t e2 = g_rccl.GroupEnd();
        if (e == NCCL_SUCCESS_) e = e2;
        if (e != NCCL_SUCCESS_) { sync_all(m); return mfail(RAT_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(e)); }
        m->n_allgathers++;
    } else if (G > 1) {
        // logical devices (or no communicator): the same data movement by device copies.  Block r is complete when r's stream reaches
        // its event; every destination stream waits for every source before it copies block r into slot r of its gathered buffer.
        for (int r = 0; r < G; ++r) {
            MHIP_SYNC(hipSetDevice(m->dev[r]));
            MHIP_SYNC(hipEventRecord(m->ev[r], (hipStream_t)rat_stream(m->h[r])));
        }
        for (int g = 0; g < G; ++g) {
            MHIP_SYNC(hipSetDevice(m->dev[g]));
            hipStream_t s = (hipStream_t)rat_stream(m->h[g]);
            for (int r = 0; r < G; ++r) {
                if (r != g) MHIP_SYNC(hipStreamWaitEvent(s, m->ev[r], 0));
                MHIP_SYNC(hipMemcpyAsync(m->d_all[g] + (size_t)r * bb, m->d_send[r], bb, hipMemcpyDeviceToDevice, s));
            }
        }
        m->n_allgathers++;
    }
    MHIP_SYNC(hipSetDevice(m->dev[0]));
    if (G > 1 || !m->comm.empty()) MHIP_SYNC(hipMemcpyAsync(p_all, m->d_all[0], bb * G, hipMemcpyDeviceToHost, (hipStream_t)rat_stream(m->h[0])));
    else MHIP_SYNC(hipMemcpyAsync(p_all, m->d_send[0], bb, hipMemcpyDeviceToHost, (hipStream_t)rat_stream(m->h[0])));   // one device, no communicator
    for (int g = 0; g < G; ++g) {                                // (every device must have finished its part of the collective)
        MHIP_SYNC(hipSetDevice(m->dev[g]));
        MHIP_SYNC(hipStreamSynchronize((hipStream_t)rat_stream(m->h[g])));
    }
    for (int g = 0; g < G; ++g) {
        const size_t nb = (size_t)(hi[(size_t)g] - lo[(size_t)g]);
        const char *blk = p_all + (size_t)g * bb;
        memcpy(cost + lo[(size_t)g], blk, sizeof(double) * nb);
        if (status) memcpy(status + lo[(size_t)g], blk + (size_t)chunk * 8, 4 * nb);
        if (iters) memcpy(iters + lo[(size_t)g], blk + (size_t)chunk * 12, 4 * nb);
        if (ls_evals) memcpy(ls_evals + lo[(size_t)g], blk + (size_t)chunk * 16, 4 * nb);
    }
    return RAT_OK;
}

extern "C" rat_rc rat_multi_ce_compute_cost_ex(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B, double kl_bound,
                                               double *cost, int32_t *status, int32_t *iters, int32_t *ls_evals) {
    return multi_batch(m, x0, u0, theta, B, kl_bound, false, cost, status, iters, ls_evals);
}
extern "C" rat_rc rat_multi_ce_compute_cost(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B, double kl_bound,
                                            double *cost) {
    return multi_batch(m, x0, u0, theta, B, kl_bound, false, cost, nullptr, nullptr, nullptr);
}

// The batch entry point of one handle (rat_ileqg_solve_batch; compute_value_worker over a batch, :144-167) over all devices: value (+Inf for
// failures), status, iterations and line-search evaluations of every sample.
extern "C" rat_rc rat_multi_ileqg_solve_batch(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B, double *value,
                                              int32_t *status, int32_t *iters, int32_t *ls_evals) {
    return multi_batch(m, x0, u0, theta, B, 0.0, true, value, status, iters, ls_evals);
}

// PETS (pets.jl:100-126): the S control samples of compute_cost split in contiguous blocks over the devices, all K stochastic rollouts of a
// sample on one device (their mean is the sample's cost, :150); every device evaluates its block concurrently and copies its costs straight
// to the host (the elite selection needs them there: no collective).  Injected noise is addressed by global sample index and the device
// generator is keyed by the global trajectory index, so the costs do not depend on the number of devices.
rat_rc rat_pets_enqueue(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K, int32_t use_true_model, const double *zn,
                        const double *zu, uint64_t seed, int64_t sample0, double *cost);

extern "C" rat_rc rat_multi_pets_problem_set(rat_multi m, const rat_gen_problem_desc *d) {
    if (!m || !d) return mfail(RAT_ERR_ARG, "null");
    for (int g = 0; g < m->G; ++g) MRC(rat_pets_problem_set(m->h[g], d));
    m->pets_n = d->lq.n; m->pets_m = d->lq.m; m->pets_N = d->lq.N;
    return RAT_OK;
}

extern "C" rat_rc rat_multi_pets_compute_cost(rat_multi m, const double *x0, const double *controls, int64_t S, int64_t K, int32_t use_true_model,
                                              const double *zn, const double *zu, uint64_t seed, double *cost) {
    if (!m || !x0 || !controls || !cost) return mfail(RAT_ERR_ARG, "null");
    if (m->pets_N <= 0) return mfail(RAT_ERR_NO_PROBLEM, "rat_multi_pets_problem_set was not called");
    if (S < 1 || K < 1) return mfail(RAT_ERR_ARG, "S, K must be positive");
    const int G = m->G;
    const int64_t n = m->pets_n, mm = m->pets_m, N = m->pets_N;
    if ((size_t)S > m->pets_cap) {                              // pinned landing zone of the per-sample costs
        if (m->h_pets) (void)hipHostFree(m->h_pets);
        m->h_pets = nullptr; m->pets_cap = 0;
        MHIP(hipHostMalloc((void **)&m->h_pets, sizeof(double) * (size_t)S, hipHostMallocDefault));
        m->pets_cap = (size_t)S;
    }
    for (int g = 0; g < G; ++g) {
        int64_t lo, hi;
        MRC(rat_shard_bounds(S, G, g, &lo, &hi));
        if (hi == lo) continue;
        MRC(rat_pets_enqueue(m->h[g], x0, controls + (size_t)lo * N * mm, hi - lo, K, use_true_model, zn ? zn + (size_t)lo * K * N * n : nullptr,
                             (zn && zu) ? zu + (size_t)lo * K * N : nullptr, seed, lo, m->h_pets + lo));
    }
    for (int g = 0; g < G; ++g) {
        MHIP(hipSetDevice(m->dev[g]));
        MHIP(hipStreamSynchronize((hipStream_t)rat_stream(m->h[g])));
    }
    memcpy(cost, m->h_pets, sizeof(double) * (size_t)S);
    return RAT_OK;
}

// step! (:252-335) with the cost evaluation on all devices; draws come from device 0's handle (rat_ce_set_stream / rat_ce_seed on
// rat_multi_handle(m, 0)), the update is host arithmetic (rat_ce_update)
extern "C" rat_rc rat_multi_ce_step(rat_multi m, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound, double *theta_out,
                                    double *cost_out) {
    if (!m || !c) return mfail(RAT_ERR_ARG, "null");
    if (c->num_samples > m->Bmax) return mfail(RAT_ERR_ARG, "num_samples exceeds max_batch of rat_create_multi");
    std::vector<double> theta((size_t)c->num_samples), cost((size_t)c->num_samples);
    MRC(rat_ce_begin_step(c));
    for (int redraws = 0;; ++redraws) {
        if (redraws > 1000) return mfail(RAT_ERR_DIVERGED, "CE redraw loop cut after 1000 redraws (reference would spin, App. B.11)");
        MRC(rat_ce_draw(m->h[0], c, theta.data()));
        MRC(rat_multi_ce_compute_cost(m, x0, u0, theta.data(), c->num_samples, kl_bound, cost.data()));
        c->n_solves += c->num_samples;
        if (redraws) c->n_redraws++;
        int32_t redraw = 0;
        MRC(rat_ce_update(c, theta.data(), cost.data(), &redraw));
        if (!redraw) break;
    }
    if (theta_out) memcpy(theta_out, theta.data(), theta.size() * 8);
    if (cost_out) memcpy(cost_out, cost.data(), cost.size() * 8);
    return RAT_OK;
}

// solve! (:364-415): CE iterations over all devices, final solve (with the theta_opt - sigma retry) on device 0
extern "C" rat_rc rat_multi_ce_solve(rat_multi m, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound, double *theta_opt,
                                     double *x, double *l, double *L, double *value, double *theta_min, double *theta_max) {
    if (!m || !c || !theta_opt || !value) return mfail(RAT_ERR_ARG, "null");
    if (!(kl_bound >= 0)) return mfail(RAT_ERR_ARG, "KL Divergence Bound must be non-negative (:368)");
    rat_ce_initialize(c);
    c->n_final_retries = 0;
    double th_opt, tmin = 0.0, tmax = 0.0;
    if (kl_bound > 0) {
        while (c->iter_current < c->iter_max) MRC(rat_multi_ce_step(m, c, x0, u0, kl_bound, nullptr, nullptr));      // :371-373
        tmin = c->theta_min; tmax = c->theta_max;
        th_opt = c->use_theta_max ? tmax : c->mu;                                                                       // :375-382
    } else {
        th_opt = 0.0;
    }
    for (int tries = 0;; ++tries) {                                                                                     // :390-414
        if (tries > 10000) return mfail(RAT_ERR_DIVERGED, "final-solve retry loop cut (reference would spin, App. B.15)");
        int32_t st = 0; double val = 0;
        MRC(rat_ileqg_solve(m->h[0], x0, u0, th_opt, x, l, L, &val, &st, nullptr, nullptr, 0, nullptr));
        if (st == RAT_ST_OK || st == RAT_ST_ITER_MAX) {
            *theta_opt = th_opt;
            if (kl_bound > 0) { *value = val + kl_bound / th_opt; if (theta_min) *theta_min = tmin; if (theta_max) *theta_max = tmax; }
            else { *value = val; if (theta_min) *theta_min = 0.0; if (theta_max) *theta_max = 0.0; }
            return RAT_OK;
        }
        th_opt = std::fmax(0.0, th_opt - c->sigma);                                                                     // :412
        c->n_final_retries++;
    }
}
