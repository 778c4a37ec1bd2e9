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
// RCCL (librccl.so, 570 MB) is loaded with dlopen the first time a rat_multi with more than one device is created: single-GPU users of
// libratilqr_hip.so never pay for it (RATILQR_MULTI_FORCE_RCCL=1 runs the collective with a one-rank communicator: test hook).
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
enum { NCCL_SUCCESS_ = 0, NCCL_FLOAT64_ = 8 };      // ncclSuccess, ncclFloat64 / ncclDouble
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
static rat_rc mfail(rat_rc rc, const std::string &m) { rat_set_error(m.c_str()); return rc; }
#define MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return mfail(RAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define MRC(expr) do { rat_rc r_ = (expr); if (r_ != RAT_OK) return r_; } while (0)

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
    std::vector<double *> d_theta, d_cost, d_all;   // per device: theta shard [chunk_max], cost shard [chunk_max], gathered [G * chunk_max]
    double *h_stage = nullptr;                      // pinned: theta (Bmax) | gathered costs (G * chunk_max)
    double *h_pets = nullptr; size_t pets_cap = 0;  // pinned: PETS per-sample costs
    int pets_n = 0, pets_m = 0, pets_N = 0;
    int64_t n_allgathers = 0;
};

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
        for (auto *v : {&m->d_theta, &m->d_cost, &m->d_all}) if (g < (int)v->size() && (*v)[g]) (void)hipFree((*v)[g]);
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
    if (n_devices > ndev) return mfail(RAT_ERR_ARG, "rat_create_multi: more devices requested than are visible (there is no CPU fallback)");
    rat_multi m = new rat_multi_s();
    m->G = n_devices; m->Bmax = max_batch; m->chunk_max = (max_batch + n_devices - 1) / n_devices;
    m->h.assign(n_devices, nullptr);
    for (int g = 0; g < n_devices; ++g) {
        const int d = devices ? devices[g] : g;
        if (d < 0 || d >= ndev) { rat_multi_destroy(m); return mfail(RAT_ERR_ARG, "rat_create_multi: bad device index"); }
        for (int q = 0; q < g; ++q) if (m->dev[q] == d) { rat_multi_destroy(m); return mfail(RAT_ERR_ARG, "rat_create_multi: device listed twice"); }
        m->dev.push_back(d);
    }
    m->d_theta.assign(n_devices, nullptr); m->d_cost.assign(n_devices, nullptr); m->d_all.assign(n_devices, nullptr);
#define MCREATE(expr) do { if ((expr) != 0) { rat_multi_destroy(m); return mfail(RAT_ERR_HIP, std::string("rat_create_multi: ") + #expr + " failed: " + rat_last_error()); } } while (0)
    for (int g = 0; g < n_devices; ++g) {
        MCREATE(rat_create(opts, m->chunk_max, spec_eps, m->dev[g], &m->h[g]));
        MCREATE((int)hipSetDevice(m->dev[g]));
        MCREATE((int)hipMalloc((void **)&m->d_theta[g], sizeof(double) * m->chunk_max));
        MCREATE((int)hipMalloc((void **)&m->d_cost[g], sizeof(double) * m->chunk_max));
        MCREATE((int)hipMalloc((void **)&m->d_all[g], sizeof(double) * m->chunk_max * n_devices));
    }
    MCREATE((int)hipHostMalloc((void **)&m->h_stage, sizeof(double) * ((size_t)max_batch + (size_t)m->chunk_max * n_devices), hipHostMallocDefault));
    const char *force = getenv("RATILQR_MULTI_FORCE_RCCL");
    if (n_devices > 1 || (force && force[0] == '1')) {
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

// compute_cost (cross_entropy_bilevel_optimization.jl:173-195) over all devices: cost_i = value_i + kl_bound / theta_i, +Inf for failures.
extern "C" rat_rc rat_multi_ce_compute_cost(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B, double kl_bound,
                                            double *cost) {
    if (!m || !theta || !cost) return mfail(RAT_ERR_ARG, "null");
    if (B < 1 || B > m->Bmax) return mfail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create_multi");
    if (x0 && u0) MRC(rat_multi_set_initial(m, x0, u0));
    const int G = m->G;
    const int64_t chunk = (B + G - 1) / G;
    double *p_theta = m->h_stage, *p_all = m->h_stage + m->Bmax;
    memcpy(p_theta, theta, sizeof(double) * B);
    const double nan = std::nan("");
    bool polled = false;
    // every device: its theta block in, its solves enqueued -- all asynchronous on the device's own stream, no host wait in this loop
    for (int g = 0; g < G; ++g) {
        int64_t lo, hi;
        MRC(rat_shard_bounds(B, G, g, &lo, &hi));
        MHIP(hipSetDevice(m->dev[g]));
        hipStream_t s = (hipStream_t)rat_stream(m->h[g]);
        if (hi - lo < chunk) {                                   // pad slots of a short (or empty) block: NaN, never read back
            std::vector<double> pad((size_t)chunk, nan);
            MHIP(hipMemcpyAsync(m->d_cost[g], pad.data(), sizeof(double) * chunk, hipMemcpyHostToDevice, s));
            MHIP(hipStreamSynchronize(s));                       // (pad is a stack-lifetime buffer; ragged batches only)
        }
        if (hi > lo) {
            MHIP(hipMemcpyAsync(m->d_theta[g], p_theta + lo, sizeof(double) * (hi - lo), hipMemcpyHostToDevice, s));
            if (rat_batch_is_single_launch(m->h[g], hi - lo)) MRC(rat_ce_compute_cost_enqueue(m->h[g], m->d_theta[g], hi - lo, kl_bound, m->d_cost[g]));
            else polled = true;
        }
    }
    if (polled) {
        // Shards on the round-based path (speculative step sizes on a shard too large for one generation of workgroups): its host loop
        // polls the device between rounds, so one calling thread would process the devices one after the other.  A helper thread per
        // such device instead, joined before the collective; their error messages come back through rc / msg.
        std::vector<std::thread> th;
        std::vector<rat_rc> rcs((size_t)G, RAT_OK);
        std::vector<std::string> msgs((size_t)G);
        for (int g = 0; g < G; ++g) {
            int64_t lo, hi;
            MRC(rat_shard_bounds(B, G, g, &lo, &hi));
            if (hi <= lo || rat_batch_is_single_launch(m->h[g], hi - lo)) continue;
            th.emplace_back([m, g, lo, hi, kl_bound, &rcs, &msgs]() {
                if (hipSetDevice(m->dev[g]) != hipSuccess) { rcs[(size_t)g] = RAT_ERR_HIP; msgs[(size_t)g] = "hipSetDevice failed in a shard thread"; return; }
                rcs[(size_t)g] = rat_ce_compute_cost_dev(m->h[g], m->d_theta[g], hi - lo, kl_bound, m->d_cost[g]);
                if (rcs[(size_t)g] != RAT_OK) msgs[(size_t)g] = rat_last_error();
            });
        }
        for (auto &t : th) t.join();
        for (int g = 0; g < G; ++g) if (rcs[(size_t)g] != RAT_OK) return mfail(rcs[(size_t)g], msgs[(size_t)g]);
    }
    if (!m->comm.empty()) {
        // ONE collective per batch: every rank contributes `chunk` doubles and receives G * chunk, ordered behind its solves on its stream
        MNCCL(g_rccl.GroupStart());
        for (int g = 0; g < G; ++g)
            MNCCL(g_rccl.AllGather(m->d_cost[g], m->d_all[g], (size_t)chunk, NCCL_FLOAT64_, m->comm[g], (hipStream_t)rat_stream(m->h[g])));
        MNCCL(g_rccl.GroupEnd());
        m->n_allgathers++;
        MHIP(hipSetDevice(m->dev[0]));
        MHIP(hipMemcpyAsync(p_all, m->d_all[0], sizeof(double) * chunk * G, hipMemcpyDeviceToHost, (hipStream_t)rat_stream(m->h[0])));
    } else {                                                     // one device, no communicator: its block is the batch
        MHIP(hipSetDevice(m->dev[0]));
        MHIP(hipMemcpyAsync(p_all, m->d_cost[0], sizeof(double) * chunk, hipMemcpyDeviceToHost, (hipStream_t)rat_stream(m->h[0])));
    }
    for (int g = 0; g < G; ++g) {                                // (every device must have finished its part of the collective)
        MHIP(hipSetDevice(m->dev[g]));
        MHIP(hipStreamSynchronize((hipStream_t)rat_stream(m->h[g])));
    }
    for (int g = 0; g < G; ++g) {
        int64_t lo, hi;
        MRC(rat_shard_bounds(B, G, g, &lo, &hi));
        memcpy(cost + lo, p_all + (size_t)g * chunk, sizeof(double) * (hi - lo));
    }
    return RAT_OK;
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
