// driver.cpp -- host side of libratilqr_hip.so: the C ABI of include/ratilqr.h.
//
// Owns the device buffers of a handle, packs problems into the padded MFMA-native tables of layout.h,
// drives the batched iLEQG state machine (solve!/step!/line_search! of ileqg.jl:494-659 for many theta
// at once, with E speculative line-search step sizes per sample) and restates the Cross-Entropy loop of
// cross_entropy_bilevel_optimization.jl:233-415 on top of it.  There is no CPU fallback: every numeric
// result is produced by the kernels of kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ratilqr.h"
#include "kernels.h"
#include "layout.h"
#include "wide.h"
#include "ce_device.h"

static thread_local std::string g_err;
static rat_rc fail(rat_rc rc, const std::string &msg) { g_err = msg; return rc; }
void rat_set_error(const char *msg) { g_err = msg; }          // (multi.cpp reports through the same rat_last_error)

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(RAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

struct EvRec { hipEvent_t a, b; int kind; int64_t ntraj; };

struct rat_handle_s {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // speculative gain sweeps run here, concurrently with the evaluation sweep
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bool dual_forced = false;        // RATILQR_DUAL was given: no automatic choice between paired and separate speculative gain sweeps
    bool dual = false;               // paired evaluation + next-gain-sweep wavefronts on the round-based path (default for E > 1; RATILQR_DUAL)
    bool speculate = false;          // opt-in (RATILQR_SPECULATE=1): measured slower than the plain order on MI355X
    bool fused = true;               // E = 1: whole solve! per sample in one persistent-wavefront launch (switch fused = 0: round-based path)
    bool fused_req = true, rounds_only = false;   // what the switches asked for (finish_switches derives `fused` / `block_mode` from them)
    int block_req = -1;
    bool wdiag = true;               // switch wdiag = 0: diagonal time-invariant W still runs the general-W arithmetic (applied by rat_problem_set)
    bool materialize = false;        // switch materialize = 1: the one-wavefront-per-sample kernel writes and reads tile records (SURVEY 8d's wording)
    std::vector<double> x0_host, u0_host;   // padded copies of what d_x0 / d_u0 hold (rat_set_initial skips identical uploads)
    int fused_occ2 = -1;             // RATILQR_FUSED_OCC2=B0: batches of at least B0 samples run the 256-register one-recursion-per-pass variant, two samples per
                                     // SIMD (0: never; -1, the default: LQ-family batches of more samples than the device has SIMDs)
    bool fused_dual = true;          // ... with policy evaluation + following gain sweep paired in one pass (RATILQR_FUSED_DUAL=0: separate)
    int block_mode = -1;             // workgroup-per-sample solve kernel (solve_block_kernel): -1 auto, 0 never, 1 whenever it is supported (RATILQR_BLOCK)
    int block_max_b = 512;           // auto, E = 1: used for batches up to this size (RATILQR_BLOCK_MAX_B)
    int n_cu = 256;                  // compute units of the device
    long long *d_xw = nullptr;       // solve_block_psw_kernel, two workgroups per sample: [Bmax][XW_STRIDE] hand-over words (kernels.hip)
    int *d_duo_count = nullptr;      // ... samples that ran as a pair so far
    unsigned xepoch = 0;             // ... launches so far (the hand-over words carry it: nothing to clear between launches)
    int *d_census = nullptr;         // solve_block_kernel's per-CU workgroup tickets (two-wave geometry: which SIMD pair a workgroup keeps)
    bool block_shape = true;         // RATILQR_BLOCK_SHAPE=0: plain two-wave workgroups, placement left to the dispatcher
    bool block_helpers = true;       // RATILQR_BLOCK_HELPERS=0: no spare linearise waves at one workgroup per CU
    bool block_acl = false;          // switch block_acl = 1: deviation-form closed-loop rollouts in the split geometry (rollacl_body; opt-in: measured
                                     // equal at 128 samples and slower at 512 -- DESIGN.md -- and not bit-identical to the other paths)
    int path_fixed = RAT_PATH_AUTO;  // rat_set_path
    bool fly_multi = true;           // ... and the rollouts of a sample's candidates share one wavefront (rollin_multi_kernel); RATILQR_FLY_MULTI=0
    bool fly = true;                 // round-based path, E > 1, LQ family: line-search candidates are evaluated without tile records in HBM
                                     // (their sweeps form the step's tile from x_t; accepted trajectories are completed on demand); RATILQR_FLY=0
    rat_ileqg_opts opts;
    OptsDev opd;
    int Bmax = 0, E = 1;
    bool have_problem = false;
    int n = 0, m = 0, N = 0;
    ProblemDev pb;
    // problems beyond the 12 + 4 tile of the MFMA kernels (wide.hip: n <= 32, m <= 32, LQ family; solve / CE / Nelder-Mead entry points)
    bool wide = false;
    WideProblemDev wpb;
    double *w_xs = nullptr, *w_us = nullptr, *w_L = nullptr, *w_dl = nullptr;
    double *w_gq = nullptr, *w_gr = nullptr, *w_gc = nullptr;      // block form: cost gradients of a trajectory, formed 16 steps per product behind its rollout
    int *w_nom = nullptr, *w_hn = nullptr;
    std::vector<void *> pb_allocs, st_allocs;
    std::vector<double> hW;          // host copy of W (col-major, N entries) for rat_approximate_model
    int W_tv = 0;
    StateDev st;
    double *d_x0 = nullptr, *d_u0 = nullptr, *d_theta = nullptr, *d_val = nullptr, *d_opout = nullptr, *d_dump = nullptr,
           *d_dlin = nullptr;
    int *d_ist = nullptr, *d_iit = nullptr, *d_ils = nullptr;
    int *h_counters = nullptr;       // pinned, [CTR_RING][2]
    double *d_hist = nullptr; int hist_dev_cap = 0;   // eps-history of single solves (rat_ileqg_solve), grown on demand
    double *h_sol = nullptr; size_t cap_sol = 0;      // pinned staging of rat_ileqg_solve's outputs (all slots of the sample, both gain halves, eps history)
    char *h_io = nullptr;            // pinned staging of the host-pointer batch entry point: theta | value | status | iters | ls_evals, [Bmax] each
    hipEvent_t round_ev[CTR_RING] = {};
    bool have_initial = false;
    // initialize!'s open-loop trajectory of the current (x_0, u_array), shared by every sample of every batch (FusedArgs.init_*)
    double *d_init_x = nullptr, *d_init_u = nullptr, *d_init_t = nullptr;
    bool init_traj_valid = false;
    bool init_lazy = true;           // switch init_lazy: the first small batch on a new (x_0, u_array) rolls initialize! out inside its own kernel
    int init_batches = 0;            // batches run on the current (x_0, u_array) so far (the first small batch rolls initialize! out in its own kernel)
    bool init_share = true;          // RATILQR_INIT_SHARE=0: every sample rolls initialize!'s trajectory out for itself (A/B and test override)
    int pred_rounds = 1;             // rounds the previous batch needed: that many are enqueued before the host first polls
    // profiling
    bool prof = false, prof_cur = false;
    unsigned prof_mask = 0xFFFFFFFFu;   // bit k: record kernel kind k
    std::vector<EvRec> evs;
    size_t ev_used = 0;
    int64_t p_launch[RAT_K_COUNT] = {0}, p_traj[RAT_K_COUNT] = {0};
    double p_ms[RAT_K_COUNT] = {0};
    // PETS
    bool have_gen = false;
    GenDev gen;
    int gn = 0, gm = 0, gN = 0;
    std::vector<void *> gen_allocs;
    double *d_pin = nullptr, *d_pzn = nullptr, *d_pzu = nullptr, *d_ptraj = nullptr, *d_pcost = nullptr;   // d_pin: x0 | padded controls
    size_t cap_pin = 0, cap_zn = 0, cap_zu = 0, cap_traj = 0, cap_cost = 0;
    double *h_pstage = nullptr; size_t cap_pstage = 0;      // pinned staging of x0 | padded controls (pets_stage_kernel reads it over the link)
    double *d_pmu = nullptr, *d_psig = nullptr; size_t cap_pmu = 0, cap_psig = 0;   // device-resident PETS loop: mu [N][m], Sigma [N][m*m]
    int *d_perr = nullptr;                                  // ... its error word (a covariance that is not positive definite)
    double *h_pzc = nullptr; size_t cap_pzc = 0;            // ... pinned: the injected control normals of a whole solve! | mu | Sigma | error word on the way back
    bool prune = true;               // round-based path, E > 1, tile-free candidates: evaluations of candidates 1 .. E-1 stop once candidate 0 is the line search's choice
    hipStream_t stream_lo = nullptr; // ... their stream: the lowest priority the device offers
    bool wide32 = true;              // every other general size (n <= 32, m <= 32): the same in block form (wide32.h)
    bool wide16 = true;              // general sizes with n <= 16, m <= 4: the sweeps of the solve kernel in registers on the matrix pipe (wide16.h)
    bool pets_device = true;                                // switch pets_device
    double *h_pcost = nullptr; size_t cap_hpcost = 0;       // pinned landing zone of the sample costs of the synchronous call (zero-copy)
    // Nelder-Mead (rat_nm_solve): costs already evaluated for this (problem, x0, u0, kl_bound) by exact theta, and the thetas of the batch
    // that ran last (its per-sample state is still on the device: the final solve is read out of it)
    std::vector<double> nm_th, nm_c, nm_last, nm_last_v;    // (nm_last_v / nm_last_st: value and status of the last batch's samples)
    std::vector<int32_t> nm_last_st;
    uint64_t nm_key = 0, problem_serial = 0;      // what the table was filled for: hash of (problem generation, options generation, x0, u0, kl_bound)
    int psweep = 0;                               // > 2: the batched sweep operators run the segment-parallel kernel with this many waves per trajectory
    int psw_hop = 120, psw_hop_e = 140, psw_comp = 125;   // its cost model (x 100, in ordinary steps): one hop (gain sweep / evaluation), one element step -- places the cuts
    bool psw_acl = true;                          // ... with its closed-loop rollouts in deviation form (rollacl_body: what block_acl is to solve_block_kernel)
    int E_req = 1;                                // the speculation width the caller asked for (rat_create's spec_eps); E is what the handle runs
    bool spec_force = false;                      // switch spec_force: run E_req speculative candidates per line-search round (the E > 1 kernels)
    bool psw_prl = true;                          // ... its closed-loop rollouts time-parallel over the four waves (kappa == 0, time-invariant cost)
    int prl_elem = 45, prl_hop = 90, prl_epi = 100;   // the cut model of that rollout, in hundredths of an ordinary step: one element step, one hop, the terminal tile
    int64_t prl_last = 0;                         // the cuts of the last launch that ran it (cut_1 | cut_2 << 16 | cut_3 << 32; 0 = did not apply)
    bool psw_duo = true;                          // ... with two workgroups (compute units) per sample while the batch leaves half the device dark
    bool block_psw = true;                        // the workgroup-per-sample solve with time-parallel sweeps for batches of <= one sample per CU (solve_block_psw_kernel)
    uint64_t opts_serial = 0;                     // bumped by everything that can change what a solve returns without a new problem: rat_set_ileqg_opts,
                                                  // rat_debug_set, rat_set_path (the reference builds a fresh ILEQGSolver from the current options per evaluation)
    int nm_depth = 3;                // switch nm_depth: 0 no speculation beyond the step's own vertices, 1 (+ carry), 2 (+ the next step's),
                                     // 3 (+ a third iteration in rat_nm_solve's first call)
    // CE randomness
    const double *z = nullptr;
    int64_t nz = 0, zpos = 0;
    bool internal_rng = false;
    uint64_t rs[4] = {0, 0, 0, 0};
    bool have_spare = false;
    double spare = 0;
    std::vector<double> zfifo;       // normals of the built-in generator drawn ahead of their use while a batch runs on the device (same sequence)
    size_t zfifo_pos = 0;
    int64_t prefill_want = 0;        // set by rat_ce_step around its batch: how many normals to have ready when the batch returns
    // device-resident Cross-Entropy loop of rat_ce_solve (ce_device.hip): state record, normal stream, theta / cost of the batch in flight
    int pets_wave16 = 1;             // switch pets_wave16: 0 PETS rollouts four per wavefront (rounds 1-3); 1 sixteen as MFMA columns, generator
                                     // wavefronts split off for small launches; 2 sixteen, never split; 3 sixteen, always split
    bool ce_device = true;           // switch ce_device = 0: the host loop (one round trip per CE iteration)
    CeDev *d_ce = nullptr, *h_ce = nullptr;          // device record, pinned host mirror
    double *d_cez = nullptr, *h_cez = nullptr; size_t cap_cez = 0;      // standard normals: pinned host buffer and its device address (read in place)
    double *d_ce_theta = nullptr, *d_ce_cost = nullptr;
};

extern "C" int32_t rat_version(void) { return RAT_VERSION; }
extern "C" const char *rat_last_error(void) { return g_err.c_str(); }

extern "C" void rat_default_ileqg_opts(rat_ileqg_opts *o) {          // ileqg.jl:191-194
    o->mu_min = 1e-6; o->delta_0 = 2.0; o->lambda = 0.5; o->d = 1e-2; o->iter_max = 100;
    o->eps_init = 1.0; o->eps_min = 1e-6; o->adaptive_eps_init = 0;
}

static bool opts_ok(const rat_ileqg_opts *o) {                       // the @assert block ileqg.jl:195-201
    return (0 < o->lambda && o->lambda < 1) && (o->d > 0) && (o->mu_min > 0) && (o->delta_0 > 0) &&
           (0 < o->eps_init && o->eps_init <= 1) && (o->eps_init > o->eps_min) && (0 < o->eps_min && o->eps_min < 1) &&
           o->iter_max >= 1;
}
static void set_opd(rat_handle h) {
    h->opd.mu_min = h->opts.mu_min; h->opd.delta_0 = h->opts.delta_0; h->opd.lambda = h->opts.lambda;
    h->opd.d = h->opts.d; h->opd.eps_init = h->opts.eps_init; h->opd.eps_min = h->opts.eps_min;
    h->opd.iter_max = (int)std::min<int64_t>(h->opts.iter_max, 1 << 30); h->opd.adaptive = h->opts.adaptive_eps_init;
}

// ---- execution switches ---------------------------------------------------------------------------------------------------------
// Every A/B and test switch of a handle lives in this ONE table.  rat_debug_set / rat_debug_get (include/ratilqr.h) reach it by key; at
// rat_create the environment variable RATILQR_<KEY IN CAPITALS> of each entry is applied through the same setter (debug_from_env: the
// only getenv of this file).  None of them changes a result except where the header says so (wdiag: another rounding order).
struct DebugSwitch { const char *key; void (*set)(rat_handle, int64_t); int64_t (*get)(rat_handle); };
static const DebugSwitch debug_switches[] = {
    {"speculate", [](rat_handle h, int64_t v) { h->speculate = (v == 1); }, [](rat_handle h) -> int64_t { return h->speculate; }},
    {"dual", [](rat_handle h, int64_t v) { h->dual = (v == 1); h->dual_forced = true; }, [](rat_handle h) -> int64_t { return h->dual; }},
    {"fused", [](rat_handle h, int64_t v) { h->fused_req = (v != 0); h->rounds_only = (v == 0); }, [](rat_handle h) -> int64_t { return h->fused; }},
    {"fused_dual", [](rat_handle, int64_t) {}, [](rat_handle h) -> int64_t { return h->fused_dual; }},   // (retired in round 6: always paired)
    {"block", [](rat_handle h, int64_t v) { h->block_req = (v == 1) ? 1 : (v == 0 ? 0 : -1); }, [](rat_handle h) -> int64_t { return h->block_mode; }},
    {"block_max_b", [](rat_handle h, int64_t v) { h->block_max_b = (int)v; }, [](rat_handle h) -> int64_t { return h->block_max_b; }},
    {"block_shape", [](rat_handle h, int64_t v) { h->block_shape = (v != 0); }, [](rat_handle h) -> int64_t { return h->block_shape; }},
    {"block_helpers", [](rat_handle h, int64_t v) { h->block_helpers = (v != 0); }, [](rat_handle h) -> int64_t { return h->block_helpers; }},
    {"block_acl", [](rat_handle h, int64_t v) { h->block_acl = (v != 0); }, [](rat_handle h) -> int64_t { return h->block_acl; }},
    {"init_share", [](rat_handle h, int64_t v) { h->init_share = (v != 0); }, [](rat_handle h) -> int64_t { return h->init_share; }},
    {"init_lazy", [](rat_handle h, int64_t v) { h->init_lazy = (v != 0); }, [](rat_handle h) -> int64_t { return h->init_lazy; }},
    {"fly", [](rat_handle h, int64_t v) { h->fly = (v != 0); }, [](rat_handle h) -> int64_t { return h->fly; }},
    {"fly_multi", [](rat_handle h, int64_t v) { h->fly_multi = (v != 0); }, [](rat_handle h) -> int64_t { return h->fly_multi; }},
    {"fused_occ2", [](rat_handle h, int64_t v) { h->fused_occ2 = (int)v; }, [](rat_handle h) -> int64_t { return h->fused_occ2; }},
    {"wide16", [](rat_handle h, int64_t v) { h->wide16 = v != 0; }, [](rat_handle h) -> int64_t {
         // EFFECTIVE: with a problem set, whether its solves run the register form (launch_wide_solve's own gate: 12 <= n <= 16, m <= 4)
         if (!h->have_problem) return h->wide16;
         return h->wide16 && h->wide && h->wpb.n >= 12 && h->wpb.n <= 16 && h->wpb.m <= 4; }},
    {"wide32", [](rat_handle h, int64_t v) { h->wide32 = v != 0; }, [](rat_handle h) -> int64_t {
         if (!h->have_problem) return h->wide32;
         const bool s16 = h->wide16 && h->wpb.n >= 12 && h->wpb.n <= 16 && h->wpb.m <= 4;
         return h->wide32 && h->wide && !s16; }},
    {"prune", [](rat_handle h, int64_t v) { h->prune = v != 0; }, [](rat_handle h) -> int64_t { return h->prune; }},
    {"wdiag", [](rat_handle h, int64_t v) { h->wdiag = (v != 0); }, [](rat_handle h) -> int64_t { return h->wdiag; }},
    {"materialize", [](rat_handle h, int64_t v) { h->materialize = (v != 0); }, [](rat_handle h) -> int64_t { return h->materialize; }},
    {"nm_depth", [](rat_handle h, int64_t v) { h->nm_depth = (v < 0 || v > 3) ? 3 : (int)v; }, [](rat_handle h) -> int64_t { return h->nm_depth; }},
    {"pets_wave16", [](rat_handle h, int64_t v) { h->pets_wave16 = (v < 0 || v > 3) ? 1 : (int)v; }, [](rat_handle h) -> int64_t { return h->pets_wave16; }},
    {"ce_device", [](rat_handle h, int64_t v) { h->ce_device = (v != 0); }, [](rat_handle h) -> int64_t { return h->ce_device; }},
    {"pets_device", [](rat_handle h, int64_t v) { h->pets_device = (v != 0); }, [](rat_handle h) -> int64_t { return h->pets_device; }},
    {"psweep", [](rat_handle h, int64_t v) { h->psweep = (v < 2) ? 0 : (int)std::min<int64_t>(v, 4); }, [](rat_handle h) -> int64_t { return h->psweep; }},
    {"psw_hop", [](rat_handle h, int64_t v) { h->psw_hop = (int)std::max<int64_t>(1, v); }, [](rat_handle h) -> int64_t { return h->psw_hop; }},
    {"psw_hop_e", [](rat_handle h, int64_t v) { h->psw_hop_e = (int)std::max<int64_t>(1, v); }, [](rat_handle h) -> int64_t { return h->psw_hop_e; }},
    {"spec_force", [](rat_handle h, int64_t v) { h->spec_force = (v != 0); }, [](rat_handle h) -> int64_t { return h->spec_force; }},
    {"spec_width", [](rat_handle, int64_t) {}, [](rat_handle h) -> int64_t { return h->E; }},
    {"block_psw", [](rat_handle h, int64_t v) { h->block_psw = (v != 0); }, [](rat_handle h) -> int64_t { return h->block_psw; }},
    {"psw_duo", [](rat_handle h, int64_t v) { h->psw_duo = (v != 0); }, [](rat_handle h) -> int64_t { return h->psw_duo; }},
    {"psw_prl", [](rat_handle h, int64_t v) { h->psw_prl = (v != 0); }, [](rat_handle h) -> int64_t { return h->psw_prl; }},
    {"prl_elem", [](rat_handle h, int64_t v) { h->prl_elem = (int)std::max<int64_t>(1, v); }, [](rat_handle h) -> int64_t { return h->prl_elem; }},
    {"prl_hop", [](rat_handle h, int64_t v) { h->prl_hop = (int)std::max<int64_t>(0, v); }, [](rat_handle h) -> int64_t { return h->prl_hop; }},
    {"prl_epi", [](rat_handle h, int64_t v) { h->prl_epi = (int)std::max<int64_t>(0, v); }, [](rat_handle h) -> int64_t { return h->prl_epi; }},
    {"prl_cuts", [](rat_handle, int64_t) {}, [](rat_handle h) -> int64_t { return h->prl_last; }},
    {"psw_duo_count", [](rat_handle h, int64_t) { if (h->d_duo_count) { (void)hipStreamSynchronize(h->stream); (void)hipMemset(h->d_duo_count, 0, sizeof(int)); } },
     [](rat_handle h) -> int64_t { int c = 0; if (h->d_duo_count) { (void)hipStreamSynchronize(h->stream); (void)hipMemcpy(&c, h->d_duo_count, sizeof(int), hipMemcpyDeviceToHost); } return c; }},
    {"psw_acl", [](rat_handle h, int64_t v) { h->psw_acl = (v != 0); }, [](rat_handle h) -> int64_t { return h->psw_acl; }},
    {"psw_comp", [](rat_handle h, int64_t v) { h->psw_comp = (int)std::max<int64_t>(100, v); }, [](rat_handle h) -> int64_t { return h->psw_comp; }},
};
// what the requests amount to on this handle (speculation width, forced pairings)
static void finish_switches(rat_handle h) {
    // spec_eps is an UPPER BOUND on the speculation width: evaluating line-search candidates ahead of the sequential rule
    // is result-identical and pays only where SIMDs would otherwise idle -- and since the time-parallel sweeps (round 5) and the second
    // compute unit per sample (round 6) the sequential rule is at least as fast at every batch size measured on this device (DESIGN.md
    // section 3).  So a handle runs its samples with E = 1 unless the switch spec_force asks for the requested width.
    h->E = (h->spec_force || h->E_req < 1) ? std::max(h->E_req, 1) : 1;
    if (!h->dual_forced) h->dual = h->E > 1;     // E > 1: candidate 0 in paired wavefronts beside the other candidates' evaluation (+3.5 % at E = 8)
    h->fused = h->fused_req && !(h->speculate || h->dual || h->E != 1);
    h->block_mode = h->rounds_only ? 0 : h->block_req;       // fused = 0 is "the round-based path": no single-launch solve at all
}
static void debug_from_env(rat_handle h) {
    for (const DebugSwitch &sw : debug_switches) {
        std::string name = "RATILQR_";
        for (const char *c = sw.key; *c; ++c) name += (char)toupper((unsigned char)*c);
        if (const char *e = getenv(name.c_str())) {
            // (historic spellings: "B0" style numbers are plain integers; "0" / "1" flags)
            char *end = nullptr;
            const long long v = strtoll(e, &end, 10);
            if (end != e) sw.set(h, (int64_t)v);
        }
    }
}

extern "C" rat_rc rat_create(const rat_ileqg_opts *opts, int32_t max_batch, int32_t spec_eps, int32_t device, rat_handle *out) {
    if (!out || max_batch < 1 || spec_eps < 1 || spec_eps > 64) return fail(RAT_ERR_ARG, "rat_create: bad batch / spec_eps");
    rat_ileqg_opts o;
    if (opts) o = *opts; else rat_default_ileqg_opts(&o);
    if (!opts_ok(&o)) return fail(RAT_ERR_ARG, "rat_create: ILEQGSolver option out of range (ileqg.jl:195-201)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(RAT_ERR_HIP, "rat_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RAT_ERR_ARG, "rat_create: bad device index");
    HIPCHK(hipSetDevice(device));
    rat_handle h = new rat_handle_s();
    h->device = device; h->opts = o; h->Bmax = max_batch; h->E_req = spec_eps; h->E = spec_eps;
    set_opd(h);
    memset(&h->st, 0, sizeof(h->st));
    memset(&h->pb, 0, sizeof(h->pb));
    // (a failure from here on must not leak the handle: rat_destroy copes with partially built ones)
#define CREATECHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rat_destroy(h); \
        return fail(RAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    CREATECHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATECHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    // (the pruned evaluations' stream: the lowest priority the device offers; without it the switch prune stays off)
    { int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&h->stream_lo, hipStreamNonBlocking, lo) != hipSuccess) {
          (void)hipGetLastError(); h->stream_lo = nullptr;
      } }
    // (ev_a / ev_b order the handle's two streams on ONE device: no system-scope fence)
    CREATECHK(hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming | hipEventDisableSystemFence));
    CREATECHK(hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming | hipEventDisableSystemFence));
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) h->n_cu = pr.multiProcessorCount; }
    // E = 1 batches beyond one sample per SIMD: two samples per SIMD in 256 registers each, one recursion per pass (solve_fused_kernel<.., OCC2>).
    // Tile-free like the paired kernel since round 5 (its rollouts fetch their operands step by step, the sweeps form their tiles from x_t): no
    // scratch, and the second wave fills the first one's dependency stalls -- 4096 samples 3.25 M solves/s against 2.74 M for the paired kernel
    // run in generations (profiles/r05_occ2.md).  The power-law family keeps its tile records and the paired kernel (-1..3 % otherwise).
    h->fused_occ2 = -1;
    h->block_max_b = 2 * h->n_cu;    // E = 1: a workgroup per sample while every sample can have two SIMDs
    // execution switches (tests, A/B tools, bench.py's contract secondary): ONE table (debug_switches), reachable through rat_debug_set and,
    // at creation, through the environment variable RATILQR_<KEY> of each entry -- the only place this library reads the environment
    debug_from_env(h);
    finish_switches(h);
    CREATECHK(hipMalloc((void **)&h->d_census, sizeof(int) * CENSUS_SLOTS));
    CREATECHK(hipMemset(h->d_census, 0, sizeof(int) * CENSUS_SLOTS));
    CREATECHK(hipHostMalloc((void **)&h->h_counters, 2 * CTR_RING * sizeof(int), hipHostMallocDefault));
    CREATECHK(hipHostMalloc((void **)&h->h_io, std::max<size_t>((size_t)max_batch * 28, 64), hipHostMallocDefault));
    for (int i = 0; i < CTR_RING; ++i) CREATECHK(hipEventCreateWithFlags(&h->round_ev[i], hipEventDisableTiming));
#undef CREATECHK
    *out = h;
    return RAT_OK;
}

static void free_list(std::vector<void *> &v) {
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

extern "C" void rat_destroy(rat_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);       // (speculative sweeps of the round-based path run there)
    free_list(h->pb_allocs);
    free_list(h->st_allocs);
    free_list(h->gen_allocs);
    for (double *q : {h->d_pin, h->d_pzn, h->d_pzu, h->d_ptraj, h->d_pcost, h->d_pmu, h->d_psig}) if (q) (void)hipFree(q);
    if (h->d_perr) (void)hipFree(h->d_perr);
    if (h->h_pzc) (void)hipHostFree(h->h_pzc);
    if (h->h_pcost) (void)hipHostFree(h->h_pcost);
    for (auto &e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (h->h_counters) (void)hipHostFree(h->h_counters);
    if (h->h_io) (void)hipHostFree(h->h_io);
    if (h->d_hist) (void)hipFree(h->d_hist);
    if (h->h_pstage) (void)hipHostFree(h->h_pstage);
    if (h->h_sol) (void)hipHostFree(h->h_sol);
    if (h->d_census) (void)hipFree(h->d_census);
    for (void *q : {(void *)h->d_ce, (void *)h->d_ce_theta, (void *)h->d_ce_cost}) if (q) (void)hipFree(q);      // (d_cez aliases the pinned h_cez)
    if (h->h_ce) (void)hipHostFree(h->h_ce);
    if (h->h_cez) (void)hipHostFree(h->h_cez);
    for (int i = 0; i < CTR_RING; ++i) if (h->round_ev[i]) (void)hipEventDestroy(h->round_ev[i]);
    if (h->ev_a) (void)hipEventDestroy(h->ev_a);
    if (h->ev_b) (void)hipEventDestroy(h->ev_b);
    if (h->stream_lo) (void)hipStreamDestroy(h->stream_lo);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" rat_rc rat_set_ileqg_opts(rat_handle h, const rat_ileqg_opts *opts) {
    if (!h || !opts) return fail(RAT_ERR_ARG, "null");
    if (!opts_ok(opts)) return fail(RAT_ERR_ARG, "ILEQGSolver option out of range (ileqg.jl:195-201)");
    h->opts = *opts;
    set_opd(h);
    h->opts_serial++;
    return RAT_OK;
}

template <class T>
static rat_rc dev_alloc(std::vector<void *> &list, T **p, size_t count) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    list.push_back(q);
    *p = (T *)q;
    return RAT_OK;
}
template <class T>
static rat_rc dev_upload(rat_handle h, std::vector<void *> &list, const T **p, const std::vector<T> &v) {
    T *q = nullptr;
    rat_rc rc = dev_alloc(list, &q, v.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    (void)h;
    return RAT_OK;
}

// LU inverse with partial pivoting (host, tiny): stands for inv(W), ileqg.jl:365
static bool host_inv(int n, const double *A, double *Ainv) {
    std::vector<double> a(A, A + n * n), b(n * n, 0.0);
    for (int i = 0; i < n; ++i) b[i + n * i] = 1.0;
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(a[k + n * k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(a[i + n * k]) > best) { best = std::fabs(a[i + n * k]); p = i; }
        if (best == 0.0 || best != best) return false;
        if (p != k) for (int j = 0; j < n; ++j) { std::swap(a[k + n * j], a[p + n * j]); std::swap(b[k + n * j], b[p + n * j]); }
        const double d = a[k + n * k];
        for (int i = k + 1; i < n; ++i) {
            const double f = a[i + n * k] / d;
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) a[i + n * j] -= f * a[k + n * j];
            for (int j = 0; j < n; ++j) b[i + n * j] -= f * b[k + n * j];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int k = n - 1; k >= 0; --k) {
            double v = b[k + n * j];
            for (int i = k + 1; i < n; ++i) v -= a[k + n * i] * Ainv[i + n * j];
            Ainv[k + n * j] = v / a[k + n * k];
        }
    return true;
}

static rat_rc alloc_state(rat_handle h) {
    free_list(h->st_allocs);
    StateDev &st = h->st;
    h->wide = false;
    const int N = h->N, E = h->E, B = h->Bmax;
    st.B = B; st.E = E; st.N = N;
    st.tile_stride = (long)N * TSTRIDE + TTERM;
    st.x_stride = (long)(N + 1) * XSTR;
    st.u_stride = (long)N * USTR;
    const size_t slots = (size_t)B * (E + 1);
    st.tile_alias = h->fused ? 1 : 0;          // (layout.h: candidates are linearised over the dead tiles of their nominal trajectory)
    rat_rc rc;
#define AL(ptr, cnt) if ((rc = dev_alloc(h->st_allocs, &(ptr), (cnt)))) return rc
    AL(st.tiles, (st.tile_alias ? (size_t)B : slots) * st.tile_stride);
    AL(st.xs, slots * st.x_stride + STG_PAD);          // (+ STG_PAD: stage_shared reads whole 64-double chunks, possibly past the last slot)
    AL(st.us, slots * st.u_stride + STG_PAD);
    st.l_half = (long)B * N * LSTR;
    st.dl_half = (long)B * N * USTR;
    AL(st.L, (size_t)2 * st.l_half + STG_PAD);
    AL(st.dl, (size_t)2 * st.dl_half + STG_PAD);
    AL(st.lsel, B); AL(st.mu_spec, B); AL(st.delta_spec, B); AL(st.spec_st, B);
    AL(st.theta, B); AL(st.mu, B); AL(st.delta, B); AL(st.value, B); AL(st.d_cur, B); AL(st.eps_init, B); AL(st.ls_eps, B);
    AL(st.status, B); AL(st.iter, B); AL(st.ls_active, B); AL(st.ls_count, B); AL(st.slot_nom, B); AL(st.n_ls, B); AL(st.hist_n, B);
    AL(st.value_c, (size_t)B * E); AL(st.d_c, (size_t)B * E); AL(st.flag_c, (size_t)B * E); AL(st.acc0, B);
    AL(st.counters, 2 * CTR_RING); AL(st.sink, (size_t)SINK_SLOTS * 64);
    AL(h->d_xw, (size_t)B * XW_STRIDE); AL(h->d_duo_count, 1);
    h->xepoch = 0;
    st.hist = nullptr; st.hist_cap = 0;
    AL(h->d_x0, XSTR); AL(h->d_u0, (size_t)N * USTR); AL(h->d_theta, B); AL(h->d_val, B);
    AL(h->d_ist, B); AL(h->d_iit, B); AL(h->d_ils, B);
    AL(h->d_opout, 2); AL(h->d_dump, (size_t)(N + 1) * DUMP_STRIDE); AL(h->d_dlin, (size_t)N * USTR);
    AL(h->d_init_x, (size_t)st.x_stride); AL(h->d_init_u, (size_t)st.u_stride); AL(h->d_init_t, (size_t)st.tile_stride);
    h->init_traj_valid = false; h->init_batches = 0;
#undef AL
    // padded lanes of the slot pools must be exact zeros
    HIPCHK(hipMemsetAsync(st.xs, 0, slots * st.x_stride * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st.us, 0, slots * st.u_stride * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_init_x, 0, (size_t)st.x_stride * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->d_init_u, 0, (size_t)st.u_stride * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st.L, 0, (size_t)2 * st.l_half * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st.dl, 0, (size_t)2 * st.dl_half * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(st.lsel, 0, (size_t)B * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(st.spec_st, 0, (size_t)B * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(st.flag_c, 0, (size_t)B * E * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(st.acc0, 0, (size_t)B * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(h->d_dump, 0, (size_t)(N + 1) * DUMP_STRIDE * sizeof(double), h->stream));      // (diagnostic builds count into it)
    HIPCHK(hipMemsetAsync(h->d_xw, 0, (size_t)B * XW_STRIDE * sizeof(long long), h->stream));
    HIPCHK(hipMemsetAsync(h->d_duo_count, 0, sizeof(int), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}

// ---- problems beyond n <= 12, m <= 4: tables at their own size for wide.hip ---------------------------
static rat_rc alloc_state_wide(rat_handle h) {
    free_list(h->st_allocs);
    h->d_xw = nullptr; h->d_duo_count = nullptr;
    memset(&h->st, 0, sizeof(h->st));
    const int N = h->N, n = h->n, m = h->m, B = h->Bmax;
    h->st.B = B; h->st.E = h->E; h->st.N = N;
    h->st.x_stride = (long)(N + 1) * n; h->st.u_stride = (long)N * m; h->st.tile_stride = 0;
    rat_rc rc;
#define AL(ptr, cnt) if ((rc = dev_alloc(h->st_allocs, &(ptr), (cnt)))) return rc
    AL(h->w_xs, (size_t)B * 2 * (N + 1) * n); AL(h->w_us, (size_t)B * 2 * N * m);
    AL(h->w_L, (size_t)B * N * n * m); AL(h->w_dl, (size_t)B * N * m); AL(h->w_nom, B); AL(h->w_hn, 1);
    AL(h->w_gq, (size_t)B * 2 * N * n); AL(h->w_gr, (size_t)B * 2 * N * m); AL(h->w_gc, (size_t)B * 2 * 64);
    AL(h->d_x0, n); AL(h->d_u0, (size_t)N * m); AL(h->d_theta, B); AL(h->d_val, B);
    AL(h->d_ist, B); AL(h->d_iit, B); AL(h->d_ils, B);
    AL(h->d_opout, 2); AL(h->d_dump, 1); AL(h->d_dlin, 1);
#undef AL
    h->wide = true;
    return RAT_OK;
}

// log|det W| by LU with partial pivoting (logdet(W M) = logdet W + logdet M, ileqg.jl:387)
static bool host_logabsdet(int n, const double *A, double *out) {
    std::vector<double> a(A, A + n * n);
    double acc = 0.0;
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(a[k + n * k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(a[i + n * k]) > best) { best = std::fabs(a[i + n * k]); p = i; }
        if (best == 0.0 || best != best) return false;
        if (p != k) for (int j = 0; j < n; ++j) std::swap(a[k + n * j], a[p + n * j]);
        acc += std::log(std::fabs(a[k + n * k]));
        for (int i = k + 1; i < n; ++i) {
            const double f = a[i + n * k] / a[k + n * k];
            for (int j = k + 1; j < n; ++j) a[i + n * j] -= f * a[k + n * j];
        }
    }
    *out = acc;
    return true;
}

static rat_rc problem_set_wide(rat_handle h, const rat_problem_desc *d) {
    const int n = d->n, m = d->m, N = d->N;
    if (!d->A || !d->B || !d->Q || !d->R || !d->P || !d->qv || !d->rv || !d->q0 || !d->Qf || !d->qvf)
        return fail(RAT_ERR_ARG, "LQ family: a table pointer is null");
    HIPCHK(hipStreamSynchronize(h->stream));
    free_list(h->pb_allocs);
    WideProblemDev wp;
    memset(&wp, 0, sizeof(wp));
    wp.n = n; wp.m = m; wp.N = N; wp.cost_tv = d->cost_tv ? 1 : 0; wp.W_tv = d->W_tv ? 1 : 0;
    wp.kappa = d->kappa; wp.q0f = d->q0f;
    const int Nc = wp.cost_tv ? N : 1, Nw = wp.W_tv ? N : 1;
    const size_t n2 = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    auto sym_upper = [](int k, const double *src, double *dst) {                        // Symmetric(X): the upper triangle rules
        for (int j = 0; j < k; ++j) for (int i = 0; i < k; ++i) dst[i + k * j] = (i <= j) ? src[i + k * j] : src[j + k * i];
    };
    std::vector<double> A(d->A, d->A + n2), B(d->B, d->B + nm), Q(Nc * n2), R(Nc * mm), P(d->P, d->P + Nc * nm);
    std::vector<double> qv(d->qv, d->qv + (size_t)Nc * n), rv(d->rv, d->rv + (size_t)Nc * m), q0(d->q0, d->q0 + Nc), Qf(n2), qvf(d->qvf, d->qvf + n);
    for (int k = 0; k < Nc; ++k) { sym_upper(n, d->Q + k * n2, &Q[k * n2]); sym_upper(m, d->R + k * mm, &R[k * mm]); }
    sym_upper(n, d->Qf, Qf.data());
    std::vector<double> W(d->W, d->W + Nw * n2), Winv(Nw * n2), ldW(Nw), wi(n2);
    for (int k = 0; k < Nw; ++k) {
        if (!host_inv(n, d->W + k * n2, wi.data()) || !host_logabsdet(n, d->W + k * n2, &ldW[k]))
            return fail(RAT_ERR_ARG, "rat_problem_set: W(k) is singular (inv(W) would throw, ileqg.jl:365)");
        sym_upper(n, wi.data(), &Winv[k * n2]);
    }
    h->hW = W; h->W_tv = wp.W_tv;
    // register images of the tables for the block form (wide32.h; layout: wide.h)
    const int NTi = (n <= 16 && m <= 16) ? 1 : 2, MTi = (m <= 16) ? 1 : 2;
    wp.img_nt = NTi; wp.img_mt = MTi;
    // element(i, jj) of a rows x cols block of RT x CT tiles; outside the matrix: `pad` on the diagonal, zero elsewhere
    auto image = [](int RT, int CT, int rows, int cols, double pad, auto element, double *dst) {
        for (int a = 0; a < RT; ++a) for (int b = 0; b < CT; ++b) for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, j = l & 15, i = 16 * a + 4 * r + g, jj = 16 * b + j;
            dst[(((size_t)a * CT + b) * 4 + r) * 64 + l] = (i < rows && jj < cols) ? element(i, jj) : ((i == jj) ? pad : 0.0);
        }
    };
    const size_t sNN = (size_t)NTi * NTi * WIDE_IMG_TILE, sMN = (size_t)MTi * NTi * WIDE_IMG_TILE, sMM = (size_t)MTi * MTi * WIDE_IMG_TILE,
                 sN1 = (size_t)NTi * WIDE_IMG_TILE, sM1 = (size_t)MTi * WIDE_IMG_TILE;
    std::vector<double> tA(sNN), tAT(sNN), tB(sMN), tBT(sMN), tQ(Nc * sNN), tP(Nc * sMN), tPT(Nc * sMN), tR(Nc * sMM), tqv(Nc * sN1), trv(Nc * sM1),
        tQf(sNN), tqvf(sN1), tWinv(Nw * sNN), tW(Nw * sNN);
    image(NTi, NTi, n, n, 0.0, [&](int i, int jj) { return A[i + n * jj]; }, tA.data());
    image(NTi, NTi, n, n, 0.0, [&](int i, int jj) { return A[jj + n * i]; }, tAT.data());
    image(NTi, MTi, n, m, 0.0, [&](int i, int jj) { return B[i + n * jj]; }, tB.data());
    image(MTi, NTi, m, n, 0.0, [&](int i, int jj) { return B[jj + n * i]; }, tBT.data());
    image(NTi, NTi, n, n, 0.0, [&](int i, int jj) { return Qf[i + n * jj]; }, tQf.data());
    image(NTi, 1, n, 1, 0.0, [&](int i, int) { return qvf[i]; }, tqvf.data());
    for (int k = 0; k < Nc; ++k) {
        const double *Qk = &Q[k * n2], *Pk = &P[k * nm], *Rk = &R[k * mm];
        image(NTi, NTi, n, n, 0.0, [&](int i, int jj) { return Qk[i + n * jj]; }, &tQ[k * sNN]);
        image(MTi, NTi, m, n, 0.0, [&](int i, int jj) { return Pk[i + m * jj]; }, &tP[k * sMN]);
        image(NTi, MTi, n, m, 0.0, [&](int i, int jj) { return Pk[jj + m * i]; }, &tPT[k * sMN]);
        image(MTi, MTi, m, m, 1.0, [&](int i, int jj) { return Rk[i + m * jj]; }, &tR[k * sMM]);
        image(NTi, 1, n, 1, 0.0, [&](int i, int) { return qv[(size_t)k * n + i]; }, &tqv[k * sN1]);
        image(MTi, 1, m, 1, 0.0, [&](int i, int) { return rv[(size_t)k * m + i]; }, &trv[k * sM1]);
    }
    for (int k = 0; k < Nw; ++k) {
        const double *Wk = &W[k * n2], *Wik = &Winv[k * n2];
        image(NTi, NTi, n, n, 1.0, [&](int i, int jj) { return Wik[i + n * jj]; }, &tWinv[k * sNN]);
        image(NTi, NTi, n, n, 0.0, [&](int i, int jj) { return Wk[i + n * jj]; }, &tW[k * sNN]);
    }
    rat_rc rc;
#define UP(field, vec) if ((rc = dev_upload(h, h->pb_allocs, &wp.field, vec))) return rc
    UP(A, A); UP(B, B); UP(Q, Q); UP(R, R); UP(P, P); UP(qv, qv); UP(rv, rv); UP(q0, q0); UP(Qf, Qf); UP(qvf, qvf);
    UP(W, W); UP(Winv, Winv); UP(ldW, ldW);
    UP(tA, tA); UP(tAT, tAT); UP(tB, tB); UP(tBT, tBT); UP(tQ, tQ); UP(tP, tP); UP(tPT, tPT); UP(tR, tR); UP(tqv, tqv); UP(trv, trv);
    UP(tQf, tQf); UP(tqvf, tqvf); UP(tWinv, tWinv); UP(tW, tW);
#undef UP
    const bool realloc_state = !h->have_problem || !h->wide || h->N != N || h->n != n || h->m != m;
    memset(&h->pb, 0, sizeof(h->pb));
    h->pb.model = d->model; h->pb.n = n; h->pb.m = m; h->pb.N = N;
    h->wpb = wp; h->n = n; h->m = m; h->N = N;
    h->have_problem = true; h->have_initial = false; h->init_traj_valid = false; h->init_batches = 0; h->problem_serial++;
    if (realloc_state && (rc = alloc_state_wide(h))) return rc;
    return RAT_OK;
}

extern "C" rat_rc rat_problem_set(rat_handle h, const rat_problem_desc *d) {
    if (!h || !d) return fail(RAT_ERR_ARG, "null");
    HIPCHK(hipSetDevice(h->device));
    const int n = d->n, m = d->m, N = d->N;
    if (n < 1 || m < 1 || N < 1) return fail(RAT_ERR_ARG, "rat_problem_set: n, m, N must be positive");
    if (n > RAT_NP || m > RAT_MP) {          // beyond the MFMA tile: the general-size solve kernel (wide.hip)
        if (n > WIDE_MAX_N || m > WIDE_MAX_M) return fail(RAT_ERR_UNSUPPORTED, "rat_problem_set: kernels are compiled for n <= 32, m <= 32");
        if (d->model != RAT_MODEL_LQ)
            return fail(RAT_ERR_UNSUPPORTED, "rat_problem_set: beyond n <= 12, m <= 4 only the LQ family is compiled");
        if (!d->W) return fail(RAT_ERR_ARG, "W missing");
        return problem_set_wide(h, d);
    }
    if (d->model != RAT_MODEL_LQ && d->model != RAT_MODEL_POWERLAW) return fail(RAT_ERR_UNSUPPORTED, "unknown model family");
    if (d->model == RAT_MODEL_POWERLAW && n != m) return fail(RAT_ERR_ARG, "power-law family needs n == m");
    if (!d->W) return fail(RAT_ERR_ARG, "W missing");
    HIPCHK(hipStreamSynchronize(h->stream));
    free_list(h->pb_allocs);
    ProblemDev pb;
    memset(&pb, 0, sizeof(pb));
    pb.model = d->model; pb.n = n; pb.m = m; pb.N = N; pb.cost_tv = d->cost_tv ? 1 : 0; pb.W_tv = d->W_tv ? 1 : 0;
    pb.q0f = d->q0f; pb.kappa = d->kappa;
    pb.pl_a = d->pl_a; pb.pl_b = d->pl_b; pb.pl_p = d->pl_p; pb.pl_pu = d->pl_pu; pb.pl_cx = d->pl_cx; pb.pl_cu = d->pl_cu; pb.pl_h = d->pl_h;
    const int Nc = pb.cost_tv ? N : 1, Nw = pb.W_tv ? N : 1;
    std::vector<double> Zt(192, 0.0), Ctab((size_t)Nc * 256, 0.0), lin((size_t)Nc * 16, 0.0), q0(Nc, 0.0), Qf(144, 0.0), qvf(16, 0.0);
    if (d->model == RAT_MODEL_LQ) {
        if (!d->A || !d->B || !d->Q || !d->R || !d->P || !d->qv || !d->rv || !d->q0 || !d->Qf || !d->qvf)
            return fail(RAT_ERR_ARG, "LQ family: a table pointer is null");
        for (int i = 0; i < n; ++i) {
            for (int jj = 0; jj < n; ++jj) Zt[i * 16 + jj] = d->A[i + n * jj];
            for (int g = 0; g < m; ++g) Zt[i * 16 + 12 + g] = d->B[i + n * g];
        }
        for (int k = 0; k < Nc; ++k) {
            double *C = &Ctab[(size_t)k * 256];
            const double *Q = d->Q + (size_t)k * n * n, *R = d->R + (size_t)k * m * m, *P = d->P + (size_t)k * m * n;
            for (int i = 0; i < n; ++i)
                for (int jj = 0; jj < n; ++jj) C[i * 16 + jj] = (i <= jj) ? Q[i + n * jj] : Q[jj + n * i];   // Symmetric(c_xx) :270
            for (int g = 0; g < m; ++g)
                for (int g2 = 0; g2 < m; ++g2) C[(12 + g) * 16 + 12 + g2] = (g <= g2) ? R[g + m * g2] : R[g2 + m * g];   // :271
            for (int g = 0; g < m; ++g)
                for (int jj = 0; jj < n; ++jj) { C[(12 + g) * 16 + jj] = P[g + m * jj]; C[jj * 16 + 12 + g] = P[g + m * jj]; }
            for (int g = m; g < RAT_MP; ++g) C[(12 + g) * 16 + 12 + g] = 1.0;
            for (int i = 0; i < n; ++i) lin[(size_t)k * 16 + i] = d->qv[(size_t)k * n + i];
            for (int g = 0; g < m; ++g) lin[(size_t)k * 16 + 12 + g] = d->rv[(size_t)k * m + g];
            q0[k] = d->q0[k];
        }
        for (int i = 0; i < n; ++i) {
            for (int jj = 0; jj < n; ++jj) Qf[i * 12 + jj] = (i <= jj) ? d->Qf[i + n * jj] : d->Qf[jj + n * i];
            qvf[i] = d->qvf[i];
        }
    }
    std::vector<double> Winv((size_t)Nw * 192, 0.0), Wp((size_t)Nw * 192, 0.0), epiv((size_t)Nw * 16, 1.0), ldw(Nw, 0.0);
    h->hW.assign(d->W, d->W + (size_t)Nw * n * n);
    h->W_tv = pb.W_tv;
    for (int k = 0; k < Nw; ++k) {
        const double *W = d->W + (size_t)k * n * n;
        std::vector<double> wi(n * n);
        if (!host_inv(n, W, wi.data())) return fail(RAT_ERR_ARG, "rat_problem_set: W(k) is singular (inv(W) would throw, ileqg.jl:365)");
        double *wo = &Winv[(size_t)k * 192], *wq = &Wp[(size_t)k * 192];
        for (int i = 0; i < RAT_NP; ++i)
            for (int jj = 0; jj < RAT_NP; ++jj) {
                if (i < n && jj < n) {
                    wo[i * 16 + jj] = (i <= jj) ? wi[i + n * jj] : wi[jj + n * i];   // Symmetric(inv(W) - ...) reads the upper triangle
                    wq[i * 16 + jj] = W[i + n * jj];
                } else if (i == jj) wo[i * 16 + jj] = 1.0;
            }
        // elimination pivots e_k of the padded inv(W): logdet(W M) = sum log(d_k / e_k)
        double a[12][12];
        for (int i = 0; i < 12; ++i) for (int jj = 0; jj < 12; ++jj) a[i][jj] = wo[i * 16 + jj];
        double pivs[12];
        for (int p = 0; p < 12; ++p) {
            const double piv = a[p][p];
            pivs[p] = piv;
            ldw[k] -= std::log(piv);
            for (int i = p + 1; i < 12; ++i) {
                const double f = a[i][p] / piv;
                for (int jj = p; jj < 12; ++jj) a[i][jj] -= f * a[p][jj];
            }
        }
        // the kernels eliminate 2x2 blocks {p, p+1}: lane p (even) multiplies det(P) by 1 / (e_p e_{p+1})
        for (int p = 0; p < 12; p += 2) { epiv[(size_t)k * 16 + p] = 1.0 / (pivs[p] * pivs[p + 1]); epiv[(size_t)k * 16 + p + 1] = 1.0; }
    }
    // time-invariant diagonal W: the sweeps fold inv(W) into the inverse of M (ProblemDev.W_diag)
    std::vector<double> Wdg(16, 1.0);
    pb.W_diag = pb.W_tv ? 0 : 1;
    if (!pb.W_tv) {
        for (int i = 0; i < n; ++i)
            for (int jj = 0; jj < n; ++jj) if (i != jj && d->W[i + n * jj] != 0.0) pb.W_diag = 0;
        for (int i = 0; i < n; ++i) Wdg[i] = Winv[i * 16 + i];
    }
    if (!h->wdiag) pb.W_diag = 0;       // debug switch wdiag = 0: the general-W arithmetic (A/B measurements, bit-identity tests)
    rat_rc rc;
#define UP(field, vec) if ((rc = dev_upload(h, h->pb_allocs, &pb.field, vec))) return rc
    UP(Wdg, Wdg);
    UP(Zt, Zt); UP(Ctab, Ctab); UP(lin, lin); UP(q0, q0); UP(Qf, Qf); UP(qvf, qvf);
    UP(Winv, Winv); UP(Wp, Wp); UP(epiv, epiv); UP(logdetW, ldw);
#undef UP
    // The slot pools rely on their padded lanes (states n..11, controls m..3 of x / u / L / dl) being exact zeros.  A problem with the
    // same N but smaller n or m would find the previous problem's values there, so the pools are rebuilt (and re-zeroed by
    // alloc_state) whenever ANY dimension changes; a problem of the same shape (a receding-horizon caller re-setting its tables every
    // control step) keeps its buffers: every live lane is rewritten by the next solve.
    const bool realloc_state = !h->have_problem || h->wide || h->N != N || h->n != n || h->m != m;
    h->pb = pb; h->n = n; h->m = m; h->N = N;
    h->have_problem = true; h->have_initial = false; h->init_traj_valid = false; h->init_batches = 0; h->problem_serial++;
    if (realloc_state && (rc = alloc_state(h))) return rc;
    return RAT_OK;
}

// ---- profiling helpers -----------------------------------------------------------------------------
static void prof_begin(rat_handle h, int kind, int64_t ntraj, hipStream_t s = nullptr) {
    h->prof_cur = h->prof && ((h->prof_mask >> kind) & 1u);
    if (!h->prof_cur) return;
    if (!s) s = h->stream;
    if (h->ev_used == h->evs.size()) {
        EvRec e;
        // (timing events: nothing on the host reads device memory through them, so no system-scope fence rides on the record)
        (void)hipEventCreateWithFlags(&e.a, hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&e.b, hipEventDisableSystemFence);
        h->evs.push_back(e);
    }
    EvRec &e = h->evs[h->ev_used];
    e.kind = kind; e.ntraj = ntraj;
    (void)hipEventRecord(e.a, s);
}
static void prof_end(rat_handle h, hipStream_t s = nullptr) {
    if (!h->prof_cur) return;
    (void)hipEventRecord(h->evs[h->ev_used].b, s ? s : h->stream);
    h->ev_used++;
}
static void prof_flush(rat_handle h) {
    if (!h->ev_used) return;
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->stream2);
    if (h->stream_lo) (void)hipStreamSynchronize(h->stream_lo);
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->evs[i].a, h->evs[i].b);
        if (h->evs[i].kind < 0) continue;                 // launches of a surplus round (no live sample)
        h->p_launch[h->evs[i].kind]++; h->p_traj[h->evs[i].kind] += h->evs[i].ntraj; h->p_ms[h->evs[i].kind] += ms;
    }
    h->ev_used = 0;
}
extern "C" rat_rc rat_profile_enable(rat_handle h, int32_t on) {
    if (!h) return RAT_ERR_ARG;
    prof_flush(h);
    h->prof = on != 0;
    h->prof_mask = (on == 1 || on == 0) ? 0xFFFFFFFFu : ((unsigned)on >> 1);   // on = 1: all kinds; else (mask << 1) | 1
    return RAT_OK;
}
extern "C" rat_rc rat_profile_reset(rat_handle h) {
    if (!h) return RAT_ERR_ARG;
    prof_flush(h);
    for (int k = 0; k < RAT_K_COUNT; ++k) { h->p_launch[k] = 0; h->p_traj[k] = 0; h->p_ms[k] = 0; }
    return RAT_OK;
}
extern "C" rat_rc rat_profile_get(rat_handle h, int64_t *launches, int64_t *traj, double *ms) {
    if (!h) return RAT_ERR_ARG;
    prof_flush(h);
    for (int k = 0; k < RAT_K_COUNT; ++k) { if (launches) launches[k] = h->p_launch[k]; if (traj) traj[k] = h->p_traj[k]; if (ms) ms[k] = h->p_ms[k]; }
    return RAT_OK;
}
#if defined(RAT_DIAG) || defined(RAT_DIAG_PHASES)
extern "C" rat_rc rat_diag_read_n(rat_handle h, double *out, int64_t off, int64_t n) {
    HIPCHK(hipMemcpy(out, h->d_dump + off, n * 8, hipMemcpyDeviceToHost));
    return RAT_OK;
}
extern "C" rat_rc rat_diag_read(rat_handle h, double *out64) {
    HIPCHK(hipMemcpy(out64, h->d_dump, 128 * 8, hipMemcpyDeviceToHost));
    return RAT_OK;
}
#endif
extern "C" void *rat_stream(rat_handle h) { return h ? (void *)h->stream : nullptr; }
extern "C" rat_rc rat_layout_info(rat_handle h, int64_t *tile_bytes, int64_t *L_bytes, int64_t *x_bytes, int64_t *u_bytes) {
    if (!h || !h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "no problem set");
    if (tile_bytes) *tile_bytes = h->st.tile_stride * 8;
    if (L_bytes) *L_bytes = (int64_t)h->N * LSTR * 8;
    if (x_bytes) *x_bytes = h->st.x_stride * 8;
    if (u_bytes) *u_bytes = h->st.u_stride * 8;
    return RAT_OK;
}

// ---- the batched solve state machine ------------------------------------------------------------------
static SweepArgs sweep_args(rat_handle h, const StateDev &st, int mode) {
    SweepArgs a;
    a.st = st; a.pb = h->pb; a.op = h->opd; a.mode = mode; a.k_first = 0; a.dl_in = nullptr; a.mu_op = 0.0; a.op_out = nullptr; a.dump = nullptr;
    a.fly = 0; a.prune = 0;
#if defined(RAT_DIAG) || defined(RAT_DIAG_PHASES)
    a.dump = h->d_dump;
#endif
    return a;
}

// Segment boundaries of the time-parallel sweep (psweep.h; tests/psweep_model.py: boundaries()).  P waves, P + 1 segments: wave P-1 runs the
// recursion over the last segment (a steps), posts, and carries on through segment P-1; wave w <= P-2 builds the element of segment w+1
// meanwhile (comp ordinary steps per step), hops when the boundary value arrives (hop steps) and runs the ordinary recursion over segment w.
// Every wave should finish together:  b_w = b_0 + w hop (wave w ends at a + (P-1-w) hop + b_w; wave P-1 at a + b_{P-1}), and the first
// element needed must be ready when the last segment is done:  a = comp b_{P-1}.
static PswCuts psweep_cuts(int N, int P, double hop, double comp) {
    PswCuts pc;
    memset(&pc, 0, sizeof(pc));
    P = std::max(1, std::min(P, PSW_MAXP));
    double b0 = 0;
    for (; P >= 2; --P) {
        b0 = ((double)N - comp * (P - 1) * hop - hop * P * (P - 1) / 2.0) / (P + comp);
        if (b0 >= 1.0 && N >= 2 * (P + 1)) break;
    }
    pc.P = P;
    if (P < 2) { pc.P = 1; pc.cut[1] = N; pc.cut[2] = N; return pc; }
    double t = 0;
    for (int w = 0; w < P; ++w) { t += b0 + w * hop; pc.cut[w + 1] = (int)(t + 0.5); }
    pc.cut[P + 1] = N;
    for (int s = 1; s <= P + 1; ++s) pc.cut[s] = std::max(pc.cut[s], pc.cut[s - 1] + 1);
    pc.cut[P + 1] = N;
    for (int s = P; s >= 1; --s) pc.cut[s] = std::min(pc.cut[s], pc.cut[s + 1] - 1);
    return pc;
}
// Segment cuts of the time-parallel rollout (kernels.hip: rollprl_body).  Wave 0 runs the ordinary rollout over n_0 steps at once; wave w >= 1
// first builds the element of segment w - 1 (e ordinary steps per step), hops (h steps; w - 1 of them are chained in front of it), then runs its
// own n_w steps.  All four finish together when  n_0 = e n_0 + n_1 = e n_1 + h + n_2 = e n_2 + 2 h + n_3  and  n_0 + ... + n_3 = N.
static PrlCuts rollprl_cuts(int N, double e, double h, double epi) {
    // Cost model in units of one ordinary rollout step: an element step costs e, a hop (box taken, Phi dx + c, box posted) h, the terminal
    // tile epi (last wave).  Wave 0 runs n_0 steps; wave w >= 1 builds the element of segment w - 1 (e n_{w-1}), takes the deviation of
    // wave w - 1 (posted at T_{w-1}; T_0 = 0), hops (T_w = max(e n_{w-1}, T_{w-1}) + h) and runs n_w steps.  All waves end together at F:
    // n_w = F - T_w (- epi for the last); F by bisection on sum n_w = N.
    auto lens = [&](double F, double *n) {
        double T = 0.0;
        n[0] = F;
        for (int w = 1; w < PRL_WAVES; ++w) {
            T = std::max(e * n[w - 1], T) + h;
            n[w] = std::max(F - T - (w == PRL_WAVES - 1 ? epi : 0.0), 1.0);
        }
        double sum = 0.0;
        for (int w = 0; w < PRL_WAVES; ++w) sum += n[w];
        return sum;
    };
    double n[PRL_WAVES], lo = 1.0, hi = (double)N;
    for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (lo + hi); (lens(mid, n) < (double)N ? lo : hi) = mid; }
    lens(hi, n);
    PrlCuts pc;
    double t = 0.0;
    pc.cut[0] = 0;
    for (int w = 0; w < PRL_WAVES; ++w) { t += n[w]; pc.cut[w + 1] = (int)(t + 0.5); }
    pc.cut[PRL_WAVES] = N;
    for (int w = 1; w <= PRL_WAVES; ++w) pc.cut[w] = std::max(pc.cut[w], pc.cut[w - 1] + 1);
    pc.cut[PRL_WAVES] = N;
    for (int w = PRL_WAVES - 1; w >= 1; --w) pc.cut[w] = std::min(pc.cut[w], pc.cut[w + 1] - 1);
    return pc;
}
// sweep launches of the batched operators: the segment-parallel kernel when the handle asks for it (switch psweep) and it covers the case
static void launch_sweep_or_psweep(rat_handle h, const SweepArgs &a, int ntraj, bool gain) {
    if (h->psweep >= 2 && psweep_supported(a, gain)) {
        const PswCuts pc = psweep_cuts(a.st.N, h->psweep, (gain ? h->psw_hop : h->psw_hop_e) / 100.0, h->psw_comp / 100.0);
        if (pc.P >= 2) { launch_psweep(a, ntraj, gain, pc, h->stream); return; }
    }
    launch_sweep(a, ntraj, gain, false, h->stream);
}

// One ROUND advances every sample by one stage of its own solve!/step!/line_search! sequence (ileqg.jl:494-659):
//   [plain gain sweep for samples that are running, not inside a line search and without a committed speculative sweep]
//   rollout + linearise (fused) of the E line-search candidates of every sample inside a line search
//   policy-evaluation sweep of the candidates   ||   on a second stream: SPECULATIVE gain sweep of the next step! on
//                                                    candidate 0's tiles (what step! would linearise if it is accepted)
//   select: replays the sequential accept rule, convergence / iter_max tests; on accepting candidate 0 it commits the
//           speculative gains (next step! begins: iter += 1, line search at eps_init) -- otherwise they are discarded.
// The speculative sweep reads exactly the inputs the reference's next solve_approximate_dp! would (tiles of the accepted
// trajectory, mu, Delta), so results are unchanged; the serial depth of a 2-iteration solve drops from 5 sweeps to 3 and
// each SIMD holds two waves whose VALU (elimination) and MFMA phases overlap.
// Samples never wait for each other: one that needs another line-search round simply gets no gain sweep this round.
// E > 1 on the round-based path: how the speculative gain sweep of the next step! (on candidate 0's trajectory) is scheduled.
//   paired  : in candidate 0's wavefront, beside its policy evaluation (sweep_dual_kernel: one pass, two recursions) -- the better use of a
//             SIMD once every SIMD has work of its own (+3.5 % at 1024 samples x 8 candidates);
//   separate: its own wavefronts on a second stream beside the evaluation sweeps of ALL candidates -- a batch that leaves SIMDs idle
//             (B (E + 1) waves within two per SIMD) pays one sweep's latency per round instead of a paired sweep's (~1.8x that).
// Same arithmetic either way (tested bit for bit).  RATILQR_SPECULATE=1 / RATILQR_DUAL=1 force one or the other.
static bool use_separate_spec(const rat_handle h, const StateDev &st) {
    if (h->speculate) return true;
    if (!h->dual || h->dual_forced || st.E == 1) return false;
    return (int64_t)st.B * (st.E + 1) <= (int64_t)8 * h->n_cu;
}

static rat_rc enqueue_round(rat_handle h, const StateDev &st, int round) {
    const int slot = round % CTR_RING;
    const int64_t nc = (int64_t)st.B * st.E;
    RolloutArgs ra; ra.st = st; ra.pb = h->pb; ra.op = h->opd; ra.dump = h->d_dump; ra.mode = 1; ra.x0 = h->d_x0; ra.u0 = h->d_u0; ra.notile = 0; ra.multi = 0;
    // no tile records for line-search candidates (LQ family, E > 1): the evaluation sweeps form each step's tile from x_t themselves, and so
    // do the gain sweeps (speculative on candidate 0, plain on whatever was accepted): nothing is completed afterwards
    const bool fly = h->fly && rollin_notile_supported(h->pb, st);
    const bool spec = use_separate_spec(h, st);
    ra.notile = fly ? 1 : 0;
    ra.multi = (fly && h->fly_multi) ? 1 : 0;
    if (h->dual && !spec) {
        // fused path (E = 1): the plain gain sweep only serves samples whose fused gain recursion was abandoned (H not PD)
        { SweepArgs sg0 = sweep_args(h, st, 0); sg0.fly = fly;       // (tile-free path: the plain gain sweep forms its tiles too -- nothing to materialise)
          prof_begin(h, RAT_K_SWEEP_GAIN, st.B); launch_sweep(sg0, st.B, true, false, h->stream); prof_end(h); }
        prof_begin(h, RAT_K_ROLLOUT, nc); launch_rollin(ra, h->stream); prof_end(h);
        // Speculation pruned (switch prune, tile-free candidates): candidate 0's paired wavefronts go first and say whether the line search
        // will settle on it (StateDev.acc0); the evaluations of candidates 1 .. E-1 -- on a stream of the lowest priority, so that the paired
        // launch, which holds a SIMD's register file alone, is dispatched ahead of them -- poll that word and stop: the sequential rule never
        // reads a candidate behind the one it accepts.  Identical outputs; the evaluations cost what the rejected candidate 0s need.
        const bool prune = h->prune && fly && st.E > 1 && h->stream_lo;
        auto launch_evals = [&](hipStream_t s2) -> rat_rc {   // candidates 1 .. E-1: plain policy evaluation, beside candidate 0's paired wavefronts
            HIPCHK(hipEventRecord(h->ev_a, h->stream));
            HIPCHK(hipStreamWaitEvent(s2, h->ev_a, 0));
            SweepArgs se = sweep_args(h, st, 1);
            se.k_first = 1;
            se.fly = fly;
            se.prune = prune ? 1 : 0;
            const int64_t n1 = (int64_t)st.B * (st.E - 1);
            prof_begin(h, RAT_K_SWEEP_EVAL, n1, s2); launch_sweep(se, (int)n1, false, false, s2); prof_end(h, s2);
            HIPCHK(hipEventRecord(h->ev_b, s2));
            return RAT_OK;
        };
        rat_rc rce;
        if (st.E > 1 && !prune && (rce = launch_evals(h->stream2))) return rce;
        if (prune) HIPCHK(hipEventRecord(h->ev_a, h->stream));               // (the rollouts' end, before the paired launch is enqueued behind it)
        { SweepArgs sd = sweep_args(h, st, 7); sd.fly = fly; sd.prune = prune ? 1 : 0;
          prof_begin(h, RAT_K_SWEEP_DUAL, st.B);
          if (prune) launch_sweep_cand0(sd, st.B, h->stream); else launch_sweep_dual(sd, st.B, h->stream);
          prof_end(h); }
        if (prune) {
            HIPCHK(hipStreamWaitEvent(h->stream_lo, h->ev_a, 0));
            SweepArgs se = sweep_args(h, st, 1);
            se.k_first = 1; se.fly = fly; se.prune = 1;
            const int64_t n1 = (int64_t)st.B * (st.E - 1);
            prof_begin(h, RAT_K_SWEEP_EVAL, n1, h->stream_lo); launch_sweep(se, (int)n1, false, false, h->stream_lo); prof_end(h, h->stream_lo);
            HIPCHK(hipEventRecord(h->ev_b, h->stream_lo));
        }
        if (st.E > 1) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_b, 0));
        prof_begin(h, RAT_K_SELECT, st.B); launch_ls_select(st, h->opd, slot, h->stream); prof_end(h);
        HIPCHK(hipMemcpyAsync(h->h_counters + 2 * slot, st.counters + 2 * slot, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->round_ev[slot], h->stream));
        return RAT_OK;
    }
    if (!spec || st.E > 1) {             // E > 1: a candidate k > 0 may be accepted, whose gain sweep was not speculated
        SweepArgs sg0 = sweep_args(h, st, 0); sg0.fly = fly;
        prof_begin(h, RAT_K_SWEEP_GAIN, st.B); launch_sweep(sg0, st.B, true, false, h->stream); prof_end(h);
    }
    prof_begin(h, RAT_K_ROLLOUT, nc); launch_rollin(ra, h->stream); prof_end(h);        // fused rollout + linearise
    if (spec) {
        HIPCHK(hipEventRecord(h->ev_a, h->stream));
        HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_a, 0));
        SweepArgs sg = sweep_args(h, st, 4); sg.fly = fly;
        prof_begin(h, RAT_K_SWEEP_GAIN, st.B, h->stream2); launch_sweep(sg, st.B, true, false, h->stream2); prof_end(h, h->stream2);
        HIPCHK(hipEventRecord(h->ev_b, h->stream2));
    }
    { SweepArgs se = sweep_args(h, st, 1); se.fly = fly;
      prof_begin(h, RAT_K_SWEEP_EVAL, nc); launch_sweep(se, (int)nc, false, false, h->stream); prof_end(h); }
    if (spec) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_b, 0));
    prof_begin(h, RAT_K_SELECT, st.B); launch_ls_select(st, h->opd, slot, h->stream); prof_end(h);
    HIPCHK(hipMemcpyAsync(h->h_counters + 2 * slot, st.counters + 2 * slot, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->round_ev[slot], h->stream));
    return RAT_OK;
}

// Which execution path runs a batch of B samples (results are bit-identical on all of them):
//   PATH_FUSED   solve_fused_kernel: one persistent wavefront per sample, evaluation + gain recursions paired in-wave  (E = 1)
//   PATH_BLOCK   solve_block_kernel: one workgroup per sample, a wave per candidate + a gain wave                      (E = 1, 2, 4, 8)
//   PATH_ROUNDS  one launch per phase, host-polled rounds                                                              (any E, operators)
// E = 1: the chip has 1024 SIMDs.  Up to 512 samples the block kernel gives every sample two SIMDs (its evaluation and gain
// recursions run side by side); from there on the in-wave pairing of the fused kernel is the better use of a SIMD (measured).
//   PATH_WIDE    wide_solve_kernel: problems beyond n <= 12, m <= 4, one workgroup per sample, whole solve! in one launch       (wide.hip)
enum Path { PATH_ROUNDS, PATH_FUSED, PATH_BLOCK, PATH_WIDE };
static Path pick_path(const rat_handle h, int B) {
    if (h->wide) return PATH_WIDE;
    if (h->path_fixed == RAT_PATH_ROUNDS) return PATH_ROUNDS;      // (rat_set_path has checked that the handle's E has the kernel)
    if (h->path_fixed == RAT_PATH_FUSED) return PATH_FUSED;
    if (h->path_fixed == RAT_PATH_BLOCK) return PATH_BLOCK;
    const bool block_ok = solve_block_supported(h->E) && h->block_mode != 0 && (h->E > 1 || h->fused) && !h->speculate &&
                          (h->E == 1 || !h->dual_forced);
    if (h->E == 1) {
        if (!h->fused) return PATH_ROUNDS;
        if (block_ok && (h->block_mode == 1 || B <= h->block_max_b)) return PATH_BLOCK;
        return PATH_FUSED;
    }
    // E > 1: a workgroup of NW waves per sample; the chip holds n_cu * (8 / NW) of them at once (two waves per SIMD).  Within one such
    // generation the block kernel wins (no launches between phases, candidates side by side on idle SIMDs); beyond it the single-wave
    // phases of a workgroup (gain sweeps) leave SIMDs idle that the round-based path fills with other samples (measured).
    const int nw = h->E == 2 ? 3 : (h->E == 4 ? 5 : 8);
    if (block_ok && (h->block_mode == 1 || B <= h->n_cu * (8 / nw))) return PATH_BLOCK;
    return PATH_ROUNDS;
}

extern "C" int32_t rat_get_path(rat_handle h, int64_t B) {
    if (!h || B < 1 || B > h->Bmax) return -1;
    switch (pick_path(h, (int)B)) {
        case PATH_ROUNDS: return RAT_PATH_ROUNDS;
        case PATH_FUSED: return RAT_PATH_FUSED;
        case PATH_BLOCK: return RAT_PATH_BLOCK;
        default: return 4;
    }
}

// The single-launch E = 1 kernels keep ONE tile bundle per sample (StateDev.tile_alias), the round-based path one per slot: moving a
// handle between them re-lays its state, so the initial trajectory has to be given again (rat_set_initial; the batch entry points that
// take x0 / u0 do it themselves).
static rat_rc alloc_state(rat_handle h);
static rat_rc relayout_if_needed(rat_handle h, bool was_alias, int was_E = -1) {
    if (was_E < 0) was_E = h->E;
    if (h->have_problem && !h->wide && (was_alias != h->fused || was_E != h->E)) {      // re-laid only when the aliasing mode or the speculation width really changes
        rat_rc rc = alloc_state(h);
        if (rc) return rc;
        h->have_initial = false; h->init_traj_valid = false; h->init_batches = 0; h->x0_host.clear(); h->u0_host.clear();
    }
    return RAT_OK;
}
extern "C" rat_rc rat_set_path(rat_handle h, int32_t path) {
    if (!h) return fail(RAT_ERR_ARG, "null");
    if (path < RAT_PATH_AUTO || path > RAT_PATH_BLOCK) return fail(RAT_ERR_ARG, "rat_set_path: unknown path");
    if (h->wide && path != RAT_PATH_AUTO) return fail(RAT_ERR_UNSUPPORTED, "rat_set_path: problems beyond n <= 12, m <= 4 run the general-size kernel only");
    if (path == RAT_PATH_FUSED && h->E != 1) return fail(RAT_ERR_UNSUPPORTED, "rat_set_path: the one-wavefront-per-sample kernel exists for spec_eps = 1 only");
    if (path == RAT_PATH_BLOCK && !solve_block_supported(h->E)) return fail(RAT_ERR_UNSUPPORTED, "rat_set_path: the workgroup-per-sample kernel exists for spec_eps 1, 2, 4, 8");
    if ((path == RAT_PATH_FUSED || path == RAT_PATH_BLOCK) && (h->speculate || (h->E == 1 && h->dual)))
        return fail(RAT_ERR_UNSUPPORTED, "rat_set_path: RATILQR_SPECULATE / RATILQR_DUAL handles run the round-based path only");
    const bool was_alias = h->st.tile_alias != 0;
    // AUTO goes back to what the handle was created with (or switched to by rat_debug_set): a handle on the round-based path by its
    // `fused = 0` switch stays there
    if (path == RAT_PATH_AUTO) finish_switches(h);
    else if (h->E == 1 && !h->speculate && !h->dual) h->fused = (path != RAT_PATH_ROUNDS);
    h->path_fixed = path;
    h->opts_serial++;
    return relayout_if_needed(h, was_alias);
}

// The one entry point behind every execution switch (tests, A/B tools, bench.py's contract secondary; include/ratilqr.h lists the keys).
extern "C" rat_rc rat_debug_set(rat_handle h, const char *key, int64_t value) {
    if (!h || !key) return fail(RAT_ERR_ARG, "null");
    for (const DebugSwitch &sw : debug_switches)
        if (!strcmp(sw.key, key)) {
            const bool was_alias = h->st.tile_alias != 0;
            const int was_E = h->E;
            sw.set(h, value);
            h->opts_serial++;
            finish_switches(h);
            if (h->path_fixed != RAT_PATH_AUTO) {            // a fixed path keeps what rat_set_path derived
                if (h->E == 1 && !h->speculate && !h->dual) h->fused = (h->path_fixed != RAT_PATH_ROUNDS);
            }
            if (h->E != was_E && h->path_fixed != RAT_PATH_AUTO && !(h->path_fixed == RAT_PATH_ROUNDS || (h->path_fixed == RAT_PATH_FUSED ? h->E == 1 : solve_block_supported(h->E))))
                h->path_fixed = RAT_PATH_AUTO;                 // (a fixed path that the new width has no kernel for)
            return relayout_if_needed(h, was_alias, was_E);
        }
    return fail(RAT_ERR_ARG, std::string("rat_debug_set: unknown switch ") + key);
}
extern "C" rat_rc rat_debug_get(rat_handle h, const char *key, int64_t *value) {
    if (!h || !key || !value) return fail(RAT_ERR_ARG, "null");
    for (const DebugSwitch &sw : debug_switches)
        if (!strcmp(sw.key, key)) { *value = sw.get(h); return RAT_OK; }
    return fail(RAT_ERR_ARG, std::string("rat_debug_get: unknown switch ") + key);
}

// outputs of a batch (device pointers, any may be null); cost = value + kl_bound / theta  (cross_entropy...jl:193)
// initialize!'s open-loop trajectory of the current (x_0, u_array), rolled out once into the handle's own slot (d_init_*): see FusedArgs.init_*
static void ensure_init_traj(rat_handle h, const RolloutArgs &ra, const double *theta_dev) {
    if (h->init_traj_valid) return;
    StateDev si = h->st;
    si.B = 1; si.xs = h->d_init_x; si.us = h->d_init_u; si.tiles = h->d_init_t;
    launch_init_state(si, h->opd, theta_dev, h->stream);     // (sample 0's control words; the batch initialises its samples again)
    RolloutArgs ri = ra; ri.st = si; ri.mode = 0; ri.notile = 0; ri.multi = 0;
    launch_rollin(ri, h->stream);
    h->init_traj_valid = true;
}

struct BatchOut { double *value = nullptr; int *status = nullptr, *iters = nullptr, *ls = nullptr; double *cost = nullptr; double kl_bound = 0.0; };

static rat_rc run_batch(rat_handle h, const double *theta_dev, int B, const BatchOut &out = BatchOut()) {
    if (!h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "rat_problem_set was not called");
    if (!h->have_initial) return fail(RAT_ERR_ARG, "rat_set_initial was not called");
    if (B < 1 || B > h->Bmax) return fail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create");
    HIPCHK(hipSetDevice(h->device));
    StateDev st = h->st;
    st.B = B;
    const Path path = pick_path(h, B);
    if (path == PATH_WIDE) {
        WideArgs wa;
        wa.pb = h->wpb; wa.op = h->opd; wa.B = B; wa.fast16 = h->wide16 ? 1 : 0; wa.fast32 = h->wide32 ? 1 : 0;
        wa.x0 = h->d_x0; wa.u0 = h->d_u0; wa.theta = theta_dev;
        wa.xs = h->w_xs; wa.us = h->w_us; wa.L = h->w_L; wa.dl = h->w_dl; wa.nom = h->w_nom;
        wa.gq = h->w_gq; wa.gr = h->w_gr; wa.gc = h->w_gc;
        wa.out_value = out.value; wa.out_status = out.status; wa.out_iters = out.iters; wa.out_ls = out.ls;
        wa.out_cost = out.cost; wa.kl_bound = out.kl_bound;
        wa.hist = h->st.hist; wa.hist_cap = h->st.hist_cap; wa.hist_n = h->w_hn;
        prof_begin(h, RAT_K_SOLVE_WIDE, B);
        HIPCHK(launch_wide_solve(wa, h->stream));
        prof_end(h);
        return RAT_OK;
    }
    if (path == PATH_ROUNDS) launch_init_state(st, h->opd, theta_dev, h->stream);     // (the single-launch solves initialise each sample themselves)
    // initialize!  (ileqg.jl:214-236): open-loop rollout, L = 0, linearise, open-loop policy evaluation; the first gain
    // sweep (step! number 1 re-linearises the same trajectory, App. B.1) runs speculatively beside it.
    RolloutArgs ra; ra.st = st; ra.pb = h->pb; ra.op = h->opd; ra.dump = h->d_dump; ra.mode = 0; ra.x0 = h->d_x0; ra.u0 = h->d_u0; ra.notile = 0; ra.multi = 0;
    if (path != PATH_ROUNDS) {       // the whole state machine below, per sample, inside one launch
        FusedArgs fa;
        fa.sw = sweep_args(h, st, 0);
        fa.sw.prune = (h->prune && st.E > 1) ? 1 : 0;       // workgroup-per-sample kernel, E > 1: candidates 1 .. E-1 stop once candidate 0 is the choice
        fa.ro = ra;
        fa.max_rounds = (int)std::min<int64_t>(((int64_t)h->opd.iter_max + 1) * 4002, 2000000000);
        fa.dual = h->fused_dual ? 1 : 0;
        fa.occ2 = (path == PATH_FUSED && h->fused_dual &&
                   (h->fused_occ2 > 0 ? B >= h->fused_occ2 : (h->fused_occ2 < 0 && h->pb.model == 1 && !h->materialize && B > 4 * (int64_t)h->n_cu))) ? 1 : 0;
        fa.mat = h->materialize ? 1 : 0;
        fa.theta_in = theta_dev;
        fa.out_value = out.value; fa.out_status = out.status; fa.out_iters = out.iters; fa.out_ls = out.ls;
        fa.out_cost = out.cost; fa.kl_bound = out.kl_bound;
        // initialize!'s rollout is the same for every sample of every batch on this (x_0, u_array): rolled out once (the per-phase kernel,
        // one wavefront) into a slot of its own; the tile-free kernels copy it instead of repeating it per sample
        fa.init_x = fa.init_u = fa.init_t = nullptr;
        // two-wave workgroups padded to one wave per SIMD (ticketed SIMD pairs): only while two workgroups per CU hold the batch -- the
        // register slots of the two waves that exit at once stay charged to the workgroup until it ends, so a third padded workgroup
        // per CU would have to wait for a whole solve (measured: 768 samples 0.515 ms padded, 0.420 ms plain)
        fa.census = (h->block_shape && h->E == 1 && B <= 2 * h->n_cu) ? h->d_census : nullptr;
        fa.helpers = (fa.census && B <= h->n_cu && h->block_helpers) ? 1 : 0;
        fa.acl = h->block_acl ? 1 : 0;
        // one sample per compute unit, LQ family: the same solve with every sweep time-parallel over the sample's four SIMDs (psweep.h)
        const bool psw = path == PATH_BLOCK && h->block_psw && fa.helpers && !h->materialize && solve_block_psw_supported(fa);
        fa.duo_stride = 0; fa.xepoch = 0; fa.xw = nullptr; fa.duo_count = nullptr;
        if (psw && h->psw_duo && 2 * ((B + 7) & ~7) <= h->n_cu) {       // half the device would be dark: two workgroups (compute units) per sample
            fa.duo_stride = (B + 7) & ~7; fa.xepoch = ++h->xepoch; fa.xw = h->d_xw; fa.duo_count = h->d_duo_count;
        }
        // ... except for the FIRST batch on a new (x_0, u_array) when a sample has a compute unit to itself: there the samples roll it out
        // themselves inside the solve (one recursion wave + three linearising waves: a few us, on compute units that would idle anyway) and the
        // 19 us launch of the shared rollout stays off the critical path of one-shot callers -- a Nelder-Mead solve!, a single rat_ileqg_solve, the
        // final solve of a fresh handle; a second batch on the same initial trajectory rolls the shared copy out for all that follow
        const bool own_init = psw && h->init_lazy && !h->init_traj_valid && h->init_batches == 0;
        if (h->init_share && h->pb.model == 1 && st.N <= ROLLIN_NST && !own_init) {
            ensure_init_traj(h, ra, theta_dev);
            fa.init_x = h->d_init_x; fa.init_u = h->d_init_u; fa.init_t = h->d_init_t;
        }
        h->init_batches++;
        fa.prl = 0;
        h->prl_last = 0;
        if (psw) {
            fa.acl = (h->psw_acl || h->block_acl) ? 1 : 0;
            if (fa.acl && h->psw_prl && h->pb.kappa == 0.0 && !h->pb.cost_tv && st.N >= 4 * PRL_WAVES) {
                fa.prl = 1;
                fa.prl_cut = rollprl_cuts(st.N, h->prl_elem / 100.0, h->prl_hop / 100.0, h->prl_epi / 100.0);
                h->prl_last = (int64_t)fa.prl_cut.cut[1] | (int64_t)fa.prl_cut.cut[2] << 16 | (int64_t)fa.prl_cut.cut[3] << 32;
            }
        }
        if (psw) {
            fa.acl = (h->psw_acl || h->block_acl) ? 1 : 0;       // (this kernel's values agree with the sequential paths to rounding anyway)
            fa.psw2e = psweep_cuts(st.N, 2, h->psw_hop_e / 100.0, h->psw_comp / 100.0);
            fa.psw2g = psweep_cuts(st.N, 2, h->psw_hop / 100.0, h->psw_comp / 100.0);
            fa.psw4e = psweep_cuts(st.N, 4, h->psw_hop_e / 100.0, h->psw_comp / 100.0);
            fa.psw4g = psweep_cuts(st.N, 4, h->psw_hop / 100.0, h->psw_comp / 100.0);
            // the four-wave gain sweep and the four-wave evaluation share one PswShared (psh[2]), whose barrier counter serves teams of ONE
            // size: the two cost models (hop 1.2 / 1.4) may settle on different team sizes at short horizons (N = 11: 3 / 2; 17, 18: 4 / 3)
            for (int it = 0; it < PSW_MAXP && fa.psw4e.P != fa.psw4g.P; ++it) {
                const int p = std::min(fa.psw4e.P, fa.psw4g.P);
                fa.psw4e = psweep_cuts(st.N, p, h->psw_hop_e / 100.0, h->psw_comp / 100.0);
                fa.psw4g = psweep_cuts(st.N, p, h->psw_hop / 100.0, h->psw_comp / 100.0);
            }
            if (fa.psw4e.P != fa.psw4g.P) fa.psw4e = fa.psw4g = psweep_cuts(st.N, 1, 1.0, 1.0);
        }
        // two waves per sample (up to two samples per CU): only the evaluation that ends the solve has an idle partner -- the two run it time-parallel
        fa.psw_last = (!psw && path == PATH_BLOCK && h->block_psw && h->E == 1 && fa.census && !fa.helpers && !h->materialize && st.N >= 8) ? 1 : 0;
        if (fa.psw_last) fa.psw2e = psweep_cuts(st.N, 2, h->psw_hop_e / 100.0, h->psw_comp / 100.0);
        prof_begin(h, path == PATH_BLOCK ? RAT_K_SOLVE_BLOCK : RAT_K_SOLVE_FUSED, B);
        if (psw) launch_solve_block_psw(fa, h->stream);
        else if (path == PATH_BLOCK) launch_solve_block(fa, h->stream); else launch_solve_fused(fa, h->stream);
        prof_end(h);
        return RAT_OK;
    }
    const bool spec = use_separate_spec(h, st);
    // tile-free speculative path with the paired first sweep: initialize!'s rollout comes from the handle's shared slot (one copy kernel
    // instead of B rollouts) and the paired sweep forms its tiles like every later sweep
    const bool share0 = h->init_share && h->dual && !spec && h->fly && rollin_notile_supported(h->pb, st);
    if (share0) {
        ensure_init_traj(h, ra, theta_dev);              // (its init_state writes to sample 0 what the batch's own has just written)
        prof_begin(h, RAT_K_ROLLOUT, B); launch_copy_initial(st, h->d_init_x, h->d_init_u, h->d_init_t, h->stream); prof_end(h);
    } else {
        prof_begin(h, RAT_K_ROLLOUT, B); launch_rollin(ra, h->stream); prof_end(h);     // fused rollout + linearise
    }
    if (h->dual && !spec) {
        SweepArgs s6 = sweep_args(h, st, 6); s6.fly = share0 ? 1 : 0;
        prof_begin(h, RAT_K_SWEEP_DUAL, B); launch_sweep_dual(s6, B, h->stream); prof_end(h);
        launch_commit_init(st, h->stream);
    } else {
    if (spec) {
        HIPCHK(hipEventRecord(h->ev_a, h->stream));
        HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_a, 0));
        prof_begin(h, RAT_K_SWEEP_GAIN, B, h->stream2); launch_sweep(sweep_args(h, st, 5), B, true, false, h->stream2); prof_end(h, h->stream2);
        HIPCHK(hipEventRecord(h->ev_b, h->stream2));
    }
    prof_begin(h, RAT_K_SWEEP_INIT, B); launch_sweep(sweep_args(h, st, 2), B, false, false, h->stream); prof_end(h);
    if (spec) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_b, 0));
        launch_commit_init(st, h->stream);
    }
    }
    // while true: step!; convergence / iter_max test   (ileqg.jl:640-654).  No host round trip per round: the host enqueues as
    // many rounds as the previous batch needed, then polls the per-round counter; it only falls back to round-by-round
    // polling for the extra rounds of a batch that runs longer (a surplus round finds no live sample and exits at once).
    const int64_t max_rounds = ((int64_t)h->opd.iter_max + 1) * (4000 / st.E + 2);
    rat_rc rc;
    const int P = std::max(1, std::min(h->pred_rounds, CTR_RING - 2));
    std::vector<size_t> ev_start;                      // first profiling event of every enqueued round
    for (int r = 0; r < P; ++r) {
        ev_start.push_back(h->ev_used);
        if ((rc = enqueue_round(h, st, r))) return rc;
    }
    int64_t r = P - 1, needed = -1;
    for (;;) {
        const int slot = (int)(r % CTR_RING);
        HIPCHK(hipEventSynchronize(h->round_ev[slot]));
        if (h->h_counters[2 * slot + 1] == 0) break;
        if (++r > max_rounds) return fail(RAT_ERR_DIVERGED, "round guard tripped");
        ev_start.push_back(h->ev_used);
        if ((rc = enqueue_round(h, st, (int)(r % (2 * CTR_RING))))) return rc;
    }
    // how many rounds were really needed (first round whose counter reads zero): prediction for the next batch
    for (int64_t q = std::max<int64_t>(0, r - (CTR_RING - 3)); q <= r; ++q)
        if (h->h_counters[2 * (q % CTR_RING) + 1] == 0) { needed = q + 1; break; }
    h->pred_rounds = (int)std::max<int64_t>(1, needed < 0 ? r + 1 : needed);
    if (needed >= 0 && (size_t)needed < ev_start.size())
        for (size_t i = ev_start[(size_t)needed]; i < h->ev_used; ++i) h->evs[i].kind = -1;   // launches of surplus rounds: not counted
    return RAT_OK;
}

extern "C" rat_rc rat_set_initial(rat_handle h, const double *x0, const double *u0) {
    if (!h || !x0 || !u0) return fail(RAT_ERR_ARG, "null");
    if (!h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "rat_problem_set was not called");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> xp(XSTR, 0.0), up((size_t)h->N * USTR, 0.0);
    if (h->wide) {                       // (own-size tables: the caller's layout as it is)
        xp.assign(x0, x0 + h->n); up.assign(u0, u0 + (size_t)h->N * h->m);
    } else {
        for (int i = 0; i < h->n; ++i) xp[i] = x0[i];
        for (int t = 0; t < h->N; ++t) for (int g = 0; g < h->m; ++g) up[(size_t)t * USTR + g] = u0[(size_t)t * h->m + g];
    }
    // (the bilevel drivers pass the same x_0 / u_array for every batch of a solve: upload only what changed)
    if (h->have_initial && xp == h->x0_host && up == h->u0_host) return RAT_OK;
    h->init_traj_valid = false; h->init_batches = 0;
    HIPCHK(hipMemcpyAsync(h->d_x0, xp.data(), xp.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_u0, up.data(), up.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->x0_host.swap(xp); h->u0_host.swap(up);
    h->have_initial = true;
    return RAT_OK;
}

extern "C" rat_rc rat_ileqg_solve_batch_dev(rat_handle h, const double *theta_dev, int64_t B, double *value_dev,
                                            int32_t *status_dev, int32_t *iters_dev, int32_t *ls_evals_dev) {
    if (!h || !theta_dev) return fail(RAT_ERR_ARG, "null");
    BatchOut out; out.value = value_dev; out.status = status_dev; out.iters = iters_dev; out.ls = ls_evals_dev;
    rat_rc rc = run_batch(h, theta_dev, (int)B, out);
    if (rc) return rc;
    if (pick_path(h, (int)B) == PATH_ROUNDS) {         // (the single-launch solves have written the outputs themselves)
        StateDev st = h->st; st.B = (int)B;
        launch_gather(st, value_dev, status_dev, iters_dev, ls_evals_dev, nullptr, 0.0, h->stream);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}

// compute_cost (cross_entropy_bilevel_optimization.jl:173-195) with theta and cost resident in HBM: cost = value + kl_bound / theta,
// +Inf for failed samples (:163-165).  One launch per batch on the fused path.
extern "C" rat_rc rat_ce_compute_cost_dev(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev) {
    if (!h || !theta_dev || !cost_dev) return fail(RAT_ERR_ARG, "null");
    BatchOut out; out.cost = cost_dev; out.kl_bound = kl_bound;
    rat_rc rc = run_batch(h, theta_dev, (int)B, out);
    if (rc) return rc;
    if (pick_path(h, (int)B) == PATH_ROUNDS) {
        StateDev st = h->st; st.B = (int)B;
        launch_gather(st, nullptr, nullptr, nullptr, nullptr, cost_dev, kl_bound, h->stream);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}

// (multi.cpp) true when a batch of B samples is ONE asynchronous launch on this handle; false when it runs the round-based path, whose
// host loop polls the device between rounds
bool rat_batch_is_single_launch(rat_handle h, int64_t B) { return h && B >= 1 && B <= h->Bmax && pick_path(h, (int)B) != PATH_ROUNDS; }

// (multi.cpp) one batch with every per-sample output left on the device: cost = value + kl_bound / theta (as_value: the value itself, what
// rat_ileqg_solve_batch returns -- not cost at kl_bound = 0, which is NaN at theta = 0), status, iterations, line-search evaluations
// (any may be null).  Single-launch paths return as soon as the launch is enqueued on the
// handle's stream unless `wait`; the round-based path, whose host loop polls the device, always returns with the batch complete.
rat_rc rat_batch_outputs_dev(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, bool as_value, double *cost_dev, int32_t *status_dev,
                             int32_t *iters_dev, int32_t *ls_dev, bool wait) {
    if (!h || !theta_dev) return fail(RAT_ERR_ARG, "null");
    if (B < 1 || B > h->Bmax) return fail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create");
    BatchOut out; out.kl_bound = kl_bound; out.status = status_dev; out.iters = iters_dev; out.ls = ls_dev;
    if (as_value) out.value = cost_dev; else out.cost = cost_dev;
    rat_rc rc = run_batch(h, theta_dev, (int)B, out);
    if (rc) return rc;
    const bool rounds = pick_path(h, (int)B) == PATH_ROUNDS;
    if (rounds) {
        StateDev st = h->st; st.B = (int)B;
        launch_gather(st, out.value, status_dev, iters_dev, ls_dev, out.cost, kl_bound, h->stream);
    }
    if (rounds || wait) HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}

extern "C" rat_rc rat_ce_compute_cost_enqueue_ex(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev,
                                                 int32_t *status_dev, int32_t *iters_dev, int32_t *ls_evals_dev) {
    if (!cost_dev) return fail(RAT_ERR_ARG, "null");
    return rat_batch_outputs_dev(h, theta_dev, B, kl_bound, false, cost_dev, status_dev, iters_dev, ls_evals_dev, false);
}

extern "C" rat_rc rat_ce_compute_cost_enqueue(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev) {
    if (!h || !theta_dev || !cost_dev) return fail(RAT_ERR_ARG, "null");
    if (B < 1 || B > h->Bmax) return fail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create");
    if (pick_path(h, (int)B) == PATH_ROUNDS) return rat_ce_compute_cost_dev(h, theta_dev, B, kl_bound, cost_dev);   // the round-based path polls its round counters
    BatchOut out; out.cost = cost_dev; out.kl_bound = kl_bound;
    return run_batch(h, theta_dev, (int)B, out);          // one launch on the handle's stream; no host wait
}

static void prefill_normals(rat_handle h, int64_t count);
extern "C" rat_rc rat_ileqg_solve_batch(rat_handle h, const double *x0, const double *u0, const double *theta, int64_t B,
                                        double *value, int32_t *status, int32_t *iters, int32_t *ls_evals) {
    if (!h || !theta || !value) return fail(RAT_ERR_ARG, "null");
    if (B < 1 || B > h->Bmax) return fail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create");
    rat_rc rc = rat_set_initial(h, x0, u0);
    if (rc) return rc;
    // theta in and the per-sample outputs back through one pinned staging area: asynchronous copies ordered on the handle's stream and
    // ONE host wait per batch (pageable-memory copies cost a staging round trip and a synchronisation each)
    const size_t M = (size_t)h->Bmax;
    double *p_theta = reinterpret_cast<double *>(h->h_io), *p_val = p_theta + M;
    int32_t *p_st = reinterpret_cast<int32_t *>(p_val + M), *p_it = p_st + M, *p_ls = p_it + M;
    memcpy(p_theta, theta, B * 8);
    // theta is read, and every per-sample output written, IN PLACE in the pinned staging area (device-visible host memory): a sample's own
    // wave reads its 8 bytes at the start of its solve and writes its 20 at the end -- no copy command before or behind the batch.
    // (The round-based path, whose several kernels re-read theta, keeps the upload.)
    const bool zc = pick_path(h, (int)B) != PATH_ROUNDS;
    if (!zc) HIPCHK(hipMemcpyAsync(h->d_theta, p_theta, B * 8, hipMemcpyHostToDevice, h->stream));
    BatchOut out;
    if (zc) { out.value = p_val; out.status = status ? p_st : nullptr; out.iters = iters ? p_it : nullptr; out.ls = ls_evals ? p_ls : nullptr; }
    else { out.value = h->d_val; out.status = h->d_ist; out.iters = h->d_iit; out.ls = h->d_ils; }
    rc = run_batch(h, zc ? p_theta : h->d_theta, (int)B, out);
    if (rc) return rc;
    if (!zc) {
        StateDev st = h->st; st.B = (int)B;
        launch_gather(st, h->d_val, h->d_ist, h->d_iit, h->d_ils, nullptr, 0.0, h->stream);
        HIPCHK(hipMemcpyAsync(p_val, h->d_val, B * 8, hipMemcpyDeviceToHost, h->stream));
        if (status) HIPCHK(hipMemcpyAsync(p_st, h->d_ist, B * 4, hipMemcpyDeviceToHost, h->stream));
        if (iters) HIPCHK(hipMemcpyAsync(p_it, h->d_iit, B * 4, hipMemcpyDeviceToHost, h->stream));
        if (ls_evals) HIPCHK(hipMemcpyAsync(p_ls, h->d_ils, B * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (h->prefill_want > 0 && pick_path(h, (int)B) != PATH_ROUNDS) prefill_normals(h, h->prefill_want);   // host work under the batch
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(value, p_val, B * 8);
    if (status) memcpy(status, p_st, B * 4);
    if (iters) memcpy(iters, p_it, B * 4);
    if (ls_evals) memcpy(ls_evals, p_ls, B * 4);
    return RAT_OK;
}

// ---- host <-> padded layout helpers -------------------------------------------------------------------
static void pad_x(const rat_handle h, const double *x, std::vector<double> &xp) {       // [n*(N+1)] -> [(N+1)*12]
    xp.assign((size_t)(h->N + 1) * XSTR, 0.0);
    for (int t = 0; t <= h->N; ++t) for (int i = 0; i < h->n; ++i) xp[(size_t)t * XSTR + i] = x[(size_t)t * h->n + i];
}
static void pad_u(const rat_handle h, const double *u, std::vector<double> &up) {
    up.assign((size_t)h->N * USTR, 0.0);
    for (int t = 0; t < h->N; ++t) for (int g = 0; g < h->m; ++g) up[(size_t)t * USTR + g] = u[(size_t)t * h->m + g];
}
static void pad_L(const rat_handle h, const double *L, std::vector<double> &Lp) {        // col-major m x n x N -> [N][4][12]
    Lp.assign((size_t)h->N * LSTR, 0.0);
    for (int t = 0; t < h->N; ++t)
        for (int g = 0; g < h->m; ++g)
            for (int jj = 0; jj < h->n; ++jj) Lp[(size_t)t * LSTR + g * 12 + jj] = L[(size_t)t * h->m * h->n + g + h->m * jj];
}
static void unpad_x(const rat_handle h, const std::vector<double> &xp, double *x) {
    for (int t = 0; t <= h->N; ++t) for (int i = 0; i < h->n; ++i) x[(size_t)t * h->n + i] = xp[(size_t)t * XSTR + i];
}
static void unpad_u(const rat_handle h, const std::vector<double> &up, double *u) {
    for (int t = 0; t < h->N; ++t) for (int g = 0; g < h->m; ++g) u[(size_t)t * h->m + g] = up[(size_t)t * USTR + g];
}
static void unpad_L(const rat_handle h, const std::vector<double> &Lp, double *L) {
    for (int t = 0; t < h->N; ++t)
        for (int g = 0; g < h->m; ++g)
            for (int jj = 0; jj < h->n; ++jj) L[(size_t)t * h->m * h->n + g + h->m * jj] = Lp[(size_t)t * LSTR + g * 12 + jj];
}

static rat_rc fetch_slot(rat_handle h, int slot, std::vector<double> *xp, std::vector<double> *up, std::vector<double> *tp) {
    const StateDev &st = h->st;
    if (xp) { xp->resize(st.x_stride); HIPCHK(hipMemcpy(xp->data(), st.xs + (size_t)slot * st.x_stride, st.x_stride * 8, hipMemcpyDeviceToHost)); }
    if (up) { up->resize(st.u_stride); HIPCHK(hipMemcpy(up->data(), st.us + (size_t)slot * st.u_stride, st.u_stride * 8, hipMemcpyDeviceToHost)); }
    if (tp) { tp->resize(st.tile_stride); HIPCHK(hipMemcpy(tp->data(), st.tiles + (size_t)tile_slot(st, slot / (st.E + 1), slot) * st.tile_stride, st.tile_stride * 8, hipMemcpyDeviceToHost)); }
    return RAT_OK;
}

// one-sample state for the operator forms: status RUNNING, slot_nom 0, given theta/mu/delta
#define WIDE_OP_MSG "this entry point is compiled for n <= 12, m <= 4"
static rat_rc op_prepare(rat_handle h, double theta, double mu, double delta, StateDev *out) {
    if (!h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "rat_problem_set was not called");
    if (h->wide) return fail(RAT_ERR_UNSUPPORTED, WIDE_OP_MSG);
    HIPCHK(hipSetDevice(h->device));
    StateDev st = h->st; st.B = 1;
    HIPCHK(hipMemcpyAsync(h->d_theta, &theta, 8, hipMemcpyHostToDevice, h->stream));
    launch_init_state(st, h->opd, h->d_theta, h->stream);
    HIPCHK(hipMemcpyAsync(st.mu, &mu, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(st.delta, &delta, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = st;
    return RAT_OK;
}

// theta_dev: the sample's theta already in device memory (the device-resident CE loop hands over &CeDev.theta_opt), else `theta` is
// uploaded.  extra_*: one more device-to-host copy enqueued before the ONE host wait of the call (the CE record).
static rat_rc ileqg_solve_impl(rat_handle h, const double *x0, const double *u0, double theta, const double *theta_dev, double *x, double *l, double *L,
                               double *value, int32_t *status, int32_t *iters, double *eps_hist, int64_t hist_cap, int64_t *hist_n,
                               void *extra_dst, const void *extra_src, size_t extra_bytes) {
    if (!h || !x0 || !u0) return fail(RAT_ERR_ARG, "null");
    rat_rc rc = rat_set_initial(h, x0, u0);
    if (rc) return rc;
    // pinned staging (h_io, >= 28 bytes): theta in, the sample's scalars back
    double *p_d = reinterpret_cast<double *>(h->h_io);            // [0] theta, then value
    int32_t *p_i = reinterpret_cast<int32_t *>(h->h_io + 16);     // status, iter, slot_nom (+ lsel, hist_n in the theta slot afterwards)
    p_d[0] = theta;
    const bool th_zc = !theta_dev && pick_path(h, 1) != PATH_ROUNDS;      // single-launch solve: theta read in place from the pinned staging word
    if (!theta_dev && !th_zc) HIPCHK(hipMemcpyAsync(h->d_theta, p_d, 8, hipMemcpyHostToDevice, h->stream));
    const int cap = (int)std::min<int64_t>(std::max<int64_t>(hist_cap, 0), 1 << 20);
    if (eps_hist && cap > 0 && cap > h->hist_dev_cap) {            // the eps-history buffer is kept (and only grown) across calls
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_hist) (void)hipFree(h->d_hist);
        h->d_hist = nullptr; h->hist_dev_cap = 0;
        HIPCHK(hipMalloc((void **)&h->d_hist, (size_t)cap * 16));
        h->hist_dev_cap = cap;
    }
    const bool want_hist = eps_hist && cap > 0;
    h->st.hist = want_hist ? h->d_hist : nullptr; h->st.hist_cap = want_hist ? cap : 0;
    // Outputs come back through ONE pinned staging area and ONE host wait: which (x, u) slot and which gain half hold the result is
    // only known when the kernel has finished, so every slot of the sample and both halves are copied (tens of KB) instead of waiting
    // for the indices first and copying afterwards (two waits and pageable-memory copies before).
    const bool wide = h->wide;
    const size_t nslot = wide ? 2 : (size_t)h->E + 1;
    const size_t xs = wide ? (size_t)(h->N + 1) * h->n : (size_t)h->st.x_stride, us = wide ? (size_t)h->N * h->m : (size_t)h->st.u_stride;
    const size_t Ls = wide ? us * h->n : (size_t)h->N * LSTR, nL = wide ? 1 : 2;
    const size_t hist_first = want_hist ? (size_t)std::min(cap, 4096) : 0;
    const size_t need = nslot * (xs + us) + nL * Ls + 2 * hist_first;
    if (need > h->cap_sol) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->h_sol) (void)hipHostFree(h->h_sol);
        h->h_sol = nullptr; h->cap_sol = 0;
        HIPCHK(hipHostMalloc((void **)&h->h_sol, need * 8, hipHostMallocDefault));
        h->cap_sol = need;
    }
    double *const s_x = h->h_sol, *const s_u = s_x + nslot * xs, *const s_L = s_u + nslot * us, *const s_h = s_L + nL * Ls;
    BatchOut wo;
    if (wide) { wo.value = h->d_val; wo.status = h->d_ist; wo.iters = h->d_iit; }
    rc = run_batch(h, theta_dev ? theta_dev : (th_zc ? p_d : h->d_theta), 1, wo);
    h->st.hist = nullptr; h->st.hist_cap = 0;
    if (rc) return rc;
    if (extra_bytes) HIPCHK(hipMemcpyAsync(extra_dst, extra_src, extra_bytes, hipMemcpyDeviceToHost, h->stream));
    int32_t *p_j = reinterpret_cast<int32_t *>(h->h_io);          // (the theta slot is free once the batch is enqueued behind its upload)
    const StateDev &st = h->st;
    int st_h, it_h, nom, lsel, hn;
    if (!wide) {
        // ONE launch packs the accepted slot, the committed gain half and the scalars straight into the pinned staging area (the device
        // writes host memory in place), then ONE host wait: no copy commands behind the solve
        double *const s_s = reinterpret_cast<double *>(h->h_io) + 2;                 // 6 doubles of scalars (h_io holds >= 64 bytes)
        launch_pack_solution(st, 0, x ? s_x : nullptr, l ? s_u : nullptr, L ? s_L : nullptr, s_s, h->stream);
        if (hist_first) HIPCHK(hipMemcpyAsync(s_h, h->d_hist, hist_first * 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        p_d[1] = s_s[0]; st_h = (int)s_s[1]; it_h = (int)s_s[2]; hn = (int)s_s[3]; nom = 0; lsel = 0;      // (the packed copies ARE the selected slot / half)
        if ((int)s_s[4] < 0 || (size_t)s_s[4] >= nslot || (int)s_s[5] < 0 || (int)s_s[5] > 1) return fail(RAT_ERR_HIP, "rat_ileqg_solve: slot index out of range");
    } else {
    HIPCHK(hipMemcpyAsync(p_d + 1, h->d_val, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(p_i + 0, h->d_ist, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(p_i + 1, h->d_iit, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(p_i + 2, h->w_nom, 4, hipMemcpyDeviceToHost, h->stream));
    if (want_hist) HIPCHK(hipMemcpyAsync(p_j + 1, h->w_hn, 4, hipMemcpyDeviceToHost, h->stream));
    if (x) HIPCHK(hipMemcpyAsync(s_x, h->w_xs, nslot * xs * 8, hipMemcpyDeviceToHost, h->stream));
    if (l) HIPCHK(hipMemcpyAsync(s_u, h->w_us, nslot * us * 8, hipMemcpyDeviceToHost, h->stream));
    if (L) HIPCHK(hipMemcpyAsync(s_L, h->w_L, Ls * 8, hipMemcpyDeviceToHost, h->stream));
    if (hist_first) HIPCHK(hipMemcpyAsync(s_h, h->d_hist, hist_first * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    st_h = p_i[0]; it_h = p_i[1]; nom = p_i[2]; lsel = 0; hn = want_hist ? p_j[1] : 0;
    }
    if (status) *status = st_h;
    if (iters) *iters = it_h;
    if (value) *value = (st_h == 0 || st_h == 3) ? p_d[1] : INFINITY;
    if (hist_n) *hist_n = hn;
    if (want_hist && hn > 0) {
        const size_t got = (size_t)std::min(hn, cap);
        memcpy(eps_hist, s_h, std::min(got, hist_first) * 16);
        if (got > hist_first)                                     // (longer histories than the speculative first block: the rest afterwards)
            HIPCHK(hipMemcpy(eps_hist + 2 * hist_first, h->d_hist + 2 * hist_first, (got - hist_first) * 16, hipMemcpyDeviceToHost));
    }
    if (nom < 0 || (size_t)nom >= nslot || lsel < 0 || lsel > 1) return fail(RAT_ERR_HIP, "rat_ileqg_solve: slot index out of range");
    if (wide) {
        if (x) memcpy(x, s_x + (size_t)nom * xs, xs * 8);
        if (l) memcpy(l, s_u + (size_t)nom * us, us * 8);
        if (L) memcpy(L, s_L, Ls * 8);
        return RAT_OK;
    }
    if (x) { std::vector<double> xp(s_x + (size_t)nom * xs, s_x + (size_t)(nom + 1) * xs); unpad_x(h, xp, x); }
    if (l) { std::vector<double> up(s_u + (size_t)nom * us, s_u + (size_t)(nom + 1) * us); unpad_u(h, up, l); }
    if (L) { std::vector<double> Lp(s_L + (size_t)lsel * Ls, s_L + (size_t)(lsel + 1) * Ls); unpad_L(h, Lp, L); }
    return RAT_OK;
}

extern "C" rat_rc rat_ileqg_solve(rat_handle h, const double *x0, const double *u0, double theta, double *x, double *l, double *L,
                                  double *value, int32_t *status, int32_t *iters, double *eps_hist, int64_t hist_cap, int64_t *hist_n) {
    return ileqg_solve_impl(h, x0, u0, theta, nullptr, x, l, L, value, status, iters, eps_hist, hist_cap, hist_n, nullptr, nullptr, 0);
}

// ---- operator forms at general size (wide.hip): the C ABI's dense column-major arrays ARE the kernel's layout ---------------------
struct WideTmp {                         // device temporaries of one operator call (operators are not a hot path)
    std::vector<void *> v;
    ~WideTmp() { for (void *p : v) (void)hipFree(p); }
    template <class T> T *out(size_t cnt) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(cnt, 1) * sizeof(T)) != hipSuccess) return nullptr;
        v.push_back(q);
        return (T *)q;
    }
    template <class T> T *in(const T *host, size_t cnt) {
        T *q = out<T>(cnt);
        if (q && hipMemcpy(q, host, cnt * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return q;
    }
};
#define WIDE_NEED(ptr) do { if (!(ptr)) return fail(RAT_ERR_HIP, "device allocation / upload failed in a general-size operator"); } while (0)
static WideOpArgs wide_op_args(rat_handle h, int opcode, long count) {
    WideOpArgs a;
    memset(&a, 0, sizeof(a));
    a.pb = h->wpb; a.op = h->opd; a.opcode = opcode; a.count = count;
    return a;
}
static rat_rc wide_run(rat_handle h, const WideOpArgs &a) {
    HIPCHK(launch_wide_op(a, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}
#define WIDE_BACK(host, dev, cnt) do { if (host) HIPCHK(hipMemcpy((host), (dev), (size_t)(cnt) * sizeof(*(host)), hipMemcpyDeviceToHost)); } while (0)

static rat_rc wide_rollout_open(rat_handle h, const double *x0, const double *u, double *x) {
    const size_t n = h->n, m = h->m, N = h->N;
    WideTmp t;
    WideOpArgs a = wide_op_args(h, WOP_ROLL_OPEN, 1);
    WIDE_NEED(a.x0 = t.in(x0, n)); WIDE_NEED(a.u = t.in(u, N * m)); WIDE_NEED(a.x_out = t.out<double>((N + 1) * n));
    rat_rc rc = wide_run(h, a);
    if (rc) return rc;
    WIDE_BACK(x, a.x_out, (N + 1) * n);
    return RAT_OK;
}
static rat_rc wide_rollout_feedback(rat_handle h, const double *xbar, const double *l, const double *L, double *x_new, double *u_new) {
    const size_t n = h->n, m = h->m, N = h->N;
    WideTmp t;
    WideOpArgs a = wide_op_args(h, WOP_ROLL_FEEDBACK, 1);
    WIDE_NEED(a.xbar = t.in(xbar, (N + 1) * n)); WIDE_NEED(a.l = t.in(l, N * m)); WIDE_NEED(a.L = t.in(L, N * m * n));
    WIDE_NEED(a.x_out = t.out<double>((N + 1) * n)); WIDE_NEED(a.u_out = t.out<double>(N * m));
    rat_rc rc = wide_run(h, a);
    if (rc) return rc;
    WIDE_BACK(x_new, a.x_out, (N + 1) * n); WIDE_BACK(u_new, a.u_out, N * m);
    return RAT_OK;
}
static rat_rc wide_integrate_cost(rat_handle h, const double *x, const double *u, double *cost) {
    const size_t n = h->n, m = h->m, N = h->N;
    WideTmp t;
    WideOpArgs a = wide_op_args(h, WOP_COST, 1);
    WIDE_NEED(a.xbar = t.in(x, (N + 1) * n)); WIDE_NEED(a.u = t.in(u, N * m)); WIDE_NEED(a.cost_out = t.out<double>(1));
    rat_rc rc = wide_run(h, a);
    if (rc) return rc;
    WIDE_BACK(cost, a.cost_out, 1);
    return RAT_OK;
}
static rat_rc wide_approximate_model(rat_handle h, const double *u, const double *x, double *q, double *qv, double *Q, double *r, double *R,
                                     double *P, double *A, double *B, double *W) {
    const size_t n = h->n, m = h->m, N = h->N;
    WideTmp t;
    WideOpArgs a = wide_op_args(h, WOP_APPROX, 1);
    WIDE_NEED(a.xbar = t.in(x, (N + 1) * n)); WIDE_NEED(a.u = t.in(u, N * m));
    WIDE_NEED(a.q = t.out<double>(N + 1)); WIDE_NEED(a.qv = t.out<double>((N + 1) * n)); WIDE_NEED(a.Q = t.out<double>((N + 1) * n * n));
    WIDE_NEED(a.r = t.out<double>(N * m)); WIDE_NEED(a.R = t.out<double>(N * m * m)); WIDE_NEED(a.P = t.out<double>(N * m * n));
    WIDE_NEED(a.A = t.out<double>(N * n * n)); WIDE_NEED(a.B = t.out<double>(N * n * m)); WIDE_NEED(a.W = t.out<double>(N * n * n));
    rat_rc rc = wide_run(h, a);
    if (rc) return rc;
    WIDE_BACK(q, a.q, N + 1); WIDE_BACK(qv, a.qv, (N + 1) * n); WIDE_BACK(Q, a.Q, (N + 1) * n * n);
    WIDE_BACK(r, a.r, N * m); WIDE_BACK(R, a.R, N * m * m); WIDE_BACK(P, a.P, N * m * n);
    WIDE_BACK(A, a.A, N * n * n); WIDE_BACK(B, a.B, N * n * m); WIDE_BACK(W, a.W, N * n * n);
    return RAT_OK;
}
// the two sweeps on caller-built tiles, B samples (B = 1: the single-sample operator forms, with the DynamicProgrammingResult dumps)
static rat_rc wide_dp(rat_handle h, bool gain, int64_t B, const double *q, const double *qv, const double *Q, const double *r, const double *R,
                      const double *P, const double *A, const double *Bm, const double *theta, double *mu, double *delta, const double *mu_in,
                      const double *Lin, const double *dlin, double *Lout, double *dlout, double *value, int32_t *status,
                      double *s, double *sv, double *S, double *g, double *G, double *H) {
    if (B < 1) return fail(RAT_ERR_ARG, "batch size must be positive");
    const size_t n = h->n, m = h->m, N = h->N, b = (size_t)B;
    WideTmp t;
    WideOpArgs a = wide_op_args(h, gain ? WOP_DP_GAIN : WOP_DP_EVAL, (long)B);
    WIDE_NEED(a.q = t.in(q, b * (N + 1))); WIDE_NEED(a.qv = t.in(qv, b * (N + 1) * n)); WIDE_NEED(a.Q = t.in(Q, b * (N + 1) * n * n));
    WIDE_NEED(a.r = t.in(r, b * N * m)); WIDE_NEED(a.R = t.in(R, b * N * m * m)); WIDE_NEED(a.P = t.in(P, b * N * m * n));
    WIDE_NEED(a.A = t.in(A, b * N * n * n)); WIDE_NEED(a.B = t.in(Bm, b * N * n * m));
    WIDE_NEED(a.theta = t.in(theta, b));
    WIDE_NEED(a.value = t.out<double>(b)); WIDE_NEED(a.status = t.out<int>(b));
    if (gain) {
        WIDE_NEED(a.mu = t.in(mu, b)); WIDE_NEED(a.delta = t.in(delta, b));
        WIDE_NEED(a.Lio = t.out<double>(b * N * m * n)); WIDE_NEED(a.dl_out = t.out<double>(b * N * m));
    } else {
        WIDE_NEED(a.mu_in = t.in(mu_in, b));
        WIDE_NEED(a.Lio = t.in(Lin, b * N * m * n));
        if (dlin) WIDE_NEED(a.dlin = t.in(dlin, b * N * m));
    }
    if (s) WIDE_NEED(a.ds = t.out<double>(N + 1));
    if (sv) WIDE_NEED(a.dsv = t.out<double>((N + 1) * n));
    if (S) WIDE_NEED(a.dS = t.out<double>((N + 1) * n * n));
    if (g) WIDE_NEED(a.dg = t.out<double>(N * m));
    if (G) WIDE_NEED(a.dG = t.out<double>(N * m * n));
    if (H) WIDE_NEED(a.dH = t.out<double>(N * m * m));
    rat_rc rc = wide_run(h, a);
    if (rc) return rc;
    if (gain) { WIDE_BACK(mu, a.mu, b); WIDE_BACK(delta, a.delta, b); WIDE_BACK(Lout, a.Lio, b * N * m * n); WIDE_BACK(dlout, a.dl_out, b * N * m); }
    WIDE_BACK(value, a.value, b);
    if (status) { std::vector<int> st(b); HIPCHK(hipMemcpy(st.data(), a.status, b * 4, hipMemcpyDeviceToHost)); for (size_t i = 0; i < b; ++i) status[i] = st[i]; }
    WIDE_BACK(s, a.ds, N + 1); WIDE_BACK(sv, a.dsv, (N + 1) * n); WIDE_BACK(S, a.dS, (N + 1) * n * n);
    WIDE_BACK(g, a.dg, N * m); WIDE_BACK(G, a.dG, N * m * n); WIDE_BACK(H, a.dH, N * m * m);
    return RAT_OK;
}

// ---- operator forms ------------------------------------------------------------------------------------
extern "C" rat_rc rat_rollout_open(rat_handle h, const double *x0, const double *u, double *x, int32_t *domain_fail) {
    if (!h || !x0 || !u || !x) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide) { if (domain_fail) *domain_fail = 0; return wide_rollout_open(h, x0, u, x); }
    rat_rc rc = rat_set_initial(h, x0, u);
    if (rc) return rc;
    StateDev st;
    if ((rc = op_prepare(h, 0.0, 0.0, h->opts.delta_0, &st))) return rc;
    RolloutArgs ra; ra.st = st; ra.pb = h->pb; ra.op = h->opd; ra.dump = h->d_dump; ra.mode = 0; ra.x0 = h->d_x0; ra.u0 = h->d_u0; ra.notile = 0; ra.multi = 0;
    launch_rollout(ra, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> xp;
    if ((rc = fetch_slot(h, 0, &xp, nullptr, nullptr))) return rc;
    unpad_x(h, xp, x);
    int st_h = 0;
    HIPCHK(hipMemcpy(&st_h, h->st.status, 4, hipMemcpyDeviceToHost));
    if (domain_fail) *domain_fail = (st_h == RAT_ST_DOMAIN);
    return RAT_OK;
}

static rat_rc put_slot0(rat_handle h, const double *x, const double *u) {
    std::vector<double> xp, up;
    if (x) { pad_x(h, x, xp); HIPCHK(hipMemcpy(h->st.xs, xp.data(), xp.size() * 8, hipMemcpyHostToDevice)); }
    if (u) { pad_u(h, u, up); HIPCHK(hipMemcpy(h->st.us, up.data(), up.size() * 8, hipMemcpyHostToDevice)); }
    return RAT_OK;
}

extern "C" rat_rc rat_rollout_feedback(rat_handle h, const double *xbar, const double *l, const double *L,
                                       double *x_new, double *u_new, int32_t *domain_fail) {
    if (!h || !xbar || !l || !L) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide) { if (domain_fail) *domain_fail = 0; return wide_rollout_feedback(h, xbar, l, L, x_new, u_new); }
    StateDev st;
    rat_rc rc = op_prepare(h, 0.0, 0.0, h->opts.delta_0, &st);
    if (rc) return rc;
    if ((rc = put_slot0(h, xbar, l))) return rc;
    std::vector<double> Lp;
    pad_L(h, L, Lp);
    HIPCHK(hipMemcpy(st.L, Lp.data(), Lp.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(st.dl, 0, (size_t)h->N * USTR * 8));
    const int one = 1;
    HIPCHK(hipMemcpy(st.ls_active, &one, 4, hipMemcpyHostToDevice));
    RolloutArgs ra; ra.st = st; ra.pb = h->pb; ra.op = h->opd; ra.dump = h->d_dump; ra.mode = 1; ra.x0 = h->d_x0; ra.u0 = h->d_u0; ra.notile = 0; ra.multi = 0;
    launch_rollout(ra, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> xp, up;
    if ((rc = fetch_slot(h, 1, &xp, &up, nullptr))) return rc;        // candidate 0 of sample 0 lives in slot 1
    if (x_new) unpad_x(h, xp, x_new);
    if (u_new) unpad_u(h, up, u_new);
    int fl = 0;
    HIPCHK(hipMemcpy(&fl, st.flag_c, 4, hipMemcpyDeviceToHost));
    if (domain_fail) *domain_fail = (fl == 2);
    return RAT_OK;
}

static bool host_chol_lower(int n, const double *A, double *Lo);

// simulate_dynamics(problem, x_0 | x_array, u_array | (l_array, L_array), rng)  -- ileqg.jl:44-55, :94-109
extern "C" rat_rc rat_rollout_noisy(rat_handle h, const double *x_nom, const double *l, const double *L, int64_t K,
                                    const double *z, uint64_t seed, double *x_out, double *u_out, double *cost_out,
                                    int32_t *domain_fail) {
    if (!h || !x_nom || !l) return fail(RAT_ERR_ARG, "null");
    if (!h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "rat_problem_set was not called");
    if (K < 1) return fail(RAT_ERR_ARG, "K must be positive");
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, Nw = h->W_tv ? N : 1;
    if (h->wide) {                       // general size: dense column-major in and out, one workgroup per rollout
        std::vector<double> Lw((size_t)Nw * n * n, 0.0);
        for (int k = 0; k < Nw; ++k)
            if (!host_chol_lower(n, h->hW.data() + (size_t)k * n * n, Lw.data() + (size_t)k * n * n))
                return fail(RAT_ERR_ARG, "W(k) is not positive definite (MvNormal would throw)");
        const size_t sn = n, sm = m, sN = N, sK = (size_t)K;
        WideTmp t;
        WideOpArgs a = wide_op_args(h, WOP_NOISY, (long)K);
        WIDE_NEED(a.xbar = t.in(x_nom, L ? (sN + 1) * sn : sn)); WIDE_NEED(a.l = t.in(l, sN * sm));
        if (L) WIDE_NEED(a.L = t.in(L, sN * sm * sn));
        if (z) WIDE_NEED(a.z = t.in(z, sK * sN * sn));
        a.seed = seed;
        WIDE_NEED(a.Wchol = t.in(Lw.data(), Lw.size()));
        if (x_out) WIDE_NEED(a.x_out = t.out<double>(sK * (sN + 1) * sn));
        if (u_out) WIDE_NEED(a.u_out = t.out<double>(sK * sN * sm));
        if (cost_out) WIDE_NEED(a.cost_out = t.out<double>(sK));
        rat_rc rcw = wide_run(h, a);
        if (rcw) return rcw;
        WIDE_BACK(x_out, a.x_out, sK * (sN + 1) * sn); WIDE_BACK(u_out, a.u_out, sK * sN * sm); WIDE_BACK(cost_out, a.cost_out, sK);
        if (domain_fail) *domain_fail = 0;
        return RAT_OK;
    }
    // lower Cholesky factors of W(k) (MvNormal sampling unwhitens with them), padded to 12 x 16 row-major
    std::vector<double> Lc((size_t)n * n), Wc((size_t)Nw * 192, 0.0);
    for (int k = 0; k < Nw; ++k) {
        if (!host_chol_lower(n, h->hW.data() + (size_t)k * n * n, Lc.data()))
            return fail(RAT_ERR_ARG, "W(k) is not positive definite (MvNormal would throw)");
        for (int i = 0; i < n; ++i) for (int jj = 0; jj <= i; ++jj) Wc[(size_t)k * 192 + i * 16 + jj] = Lc[i + n * jj];
    }
    std::vector<double> xp, up, Lp;
    if (L) { pad_x(h, x_nom, xp); pad_L(h, L, Lp); }
    else { xp.assign((size_t)(N + 1) * XSTR, 0.0); for (int i = 0; i < n; ++i) xp[i] = x_nom[i]; }
    pad_u(h, l, up);
    const int64_t chunk = std::min<int64_t>(K, 1 << 16);
    double *d_wc = nullptr, *d_x = nullptr, *d_l = nullptr, *d_L = nullptr, *d_z = nullptr, *d_xo = nullptr, *d_uo = nullptr, *d_c = nullptr;
    int *d_dom = nullptr;
    auto freeall = [&]() { for (void *q : {(void *)d_wc, (void *)d_x, (void *)d_l, (void *)d_L, (void *)d_z, (void *)d_xo, (void *)d_uo, (void *)d_c, (void *)d_dom}) if (q) (void)hipFree(q); };
#define NALLOC(ptr, count) do { if (hipMalloc((void **)&(ptr), (size_t)(count) * sizeof(*(ptr))) != hipSuccess) { freeall(); return fail(RAT_ERR_HIP, "hipMalloc failed"); } } while (0)
    NALLOC(d_wc, Wc.size()); NALLOC(d_x, xp.size()); NALLOC(d_l, up.size());
    if (L) NALLOC(d_L, Lp.size());
    if (z) NALLOC(d_z, (size_t)chunk * N * n);
    if (x_out) NALLOC(d_xo, (size_t)chunk * (N + 1) * XSTR);
    if (u_out) NALLOC(d_uo, (size_t)chunk * N * USTR);
    NALLOC(d_c, chunk); NALLOC(d_dom, chunk);
#undef NALLOC
    rat_rc rc = RAT_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == RAT_OK) rc = fail(RAT_ERR_HIP, hipGetErrorString(e)); };
    chk(hipMemcpy(d_wc, Wc.data(), Wc.size() * 8, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_x, xp.data(), xp.size() * 8, hipMemcpyHostToDevice));
    chk(hipMemcpy(d_l, up.data(), up.size() * 8, hipMemcpyHostToDevice));
    if (L) chk(hipMemcpy(d_L, Lp.data(), Lp.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> xo, uo, co((size_t)chunk);
    std::vector<int> dm((size_t)chunk);
    int any_dom = 0;
    for (int64_t k0 = 0; k0 < K && rc == RAT_OK; k0 += chunk) {
        const int64_t kc = std::min(chunk, K - k0);
        if (z) chk(hipMemcpy(d_z, z + (size_t)k0 * N * n, (size_t)kc * N * n * 8, hipMemcpyHostToDevice));
        NoisyArgs a;
        a.pb = h->pb; a.Wchol = d_wc; a.xnom = d_x; a.l = d_l; a.L = d_L; a.K = (long)kc; a.z = d_z;
        a.seed = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(k0 / chunk);       // distinct Philox keys per chunk
        a.x_out = d_xo; a.u_out = d_uo; a.cost = d_c; a.dom = d_dom;
        launch_noisy_rollout(a, h->stream);
        chk(hipStreamSynchronize(h->stream));
        chk(hipMemcpy(co.data(), d_c, (size_t)kc * 8, hipMemcpyDeviceToHost));
        chk(hipMemcpy(dm.data(), d_dom, (size_t)kc * 4, hipMemcpyDeviceToHost));
        for (int64_t q = 0; q < kc; ++q) {
            any_dom |= dm[(size_t)q];
            if (cost_out) cost_out[k0 + q] = dm[(size_t)q] ? NAN : co[(size_t)q];
        }
        if (x_out) {
            xo.resize((size_t)kc * (N + 1) * XSTR);
            chk(hipMemcpy(xo.data(), d_xo, xo.size() * 8, hipMemcpyDeviceToHost));
            for (int64_t q = 0; q < kc; ++q) for (int t = 0; t <= N; ++t) for (int i = 0; i < n; ++i)
                x_out[((size_t)(k0 + q) * (N + 1) + t) * n + i] = xo[((size_t)q * (N + 1) + t) * XSTR + i];
        }
        if (u_out) {
            uo.resize((size_t)kc * N * USTR);
            chk(hipMemcpy(uo.data(), d_uo, uo.size() * 8, hipMemcpyDeviceToHost));
            for (int64_t q = 0; q < kc; ++q) for (int t = 0; t < N; ++t) for (int g = 0; g < m; ++g)
                u_out[((size_t)(k0 + q) * N + t) * m + g] = uo[((size_t)q * N + t) * USTR + g];
        }
    }
    freeall();
    if (domain_fail) *domain_fail = any_dom;
    return rc;
}

static rat_rc linearize_slot0(rat_handle h, const double *u, const double *x, std::vector<double> *tiles, int32_t *domain_fail) {
    StateDev st;
    rat_rc rc = op_prepare(h, 0.0, 0.0, h->opts.delta_0, &st);
    if (rc) return rc;
    if ((rc = put_slot0(h, x, u))) return rc;
    LinArgs la; la.st = st; la.pb = h->pb; la.mode = 0;
    launch_linearize(la, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((rc = fetch_slot(h, 0, nullptr, nullptr, tiles))) return rc;
    int st_h = 0;
    HIPCHK(hipMemcpy(&st_h, h->st.status, 4, hipMemcpyDeviceToHost));
    if (domain_fail) *domain_fail = (st_h == RAT_ST_DOMAIN);
    return RAT_OK;
}

extern "C" rat_rc rat_integrate_cost(rat_handle h, const double *x, const double *u, double *cost) {
    if (!h || !x || !u || !cost) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide) return wide_integrate_cost(h, x, u, cost);
    std::vector<double> tp;
    int32_t dom = 0;
    rat_rc rc = linearize_slot0(h, u, x, &tp, &dom);
    if (rc) return rc;
    double acc = 0.0;                                                   // ileqg.jl:118-123
    for (int t = 0; t < h->N; ++t) acc += tp[(size_t)t * TSTRIDE + TS_q];
    acc += tp[(size_t)h->N * TSTRIDE + TT_q];
    *cost = dom ? NAN : acc;
    return RAT_OK;
}

extern "C" rat_rc rat_approximate_model(rat_handle h, const double *u, const double *x, double *q, double *qv, double *Q,
                                        double *r, double *R, double *P, double *A, double *B, double *W, int32_t *domain_fail) {
    if (!h || !x || !u) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide) { if (domain_fail) *domain_fail = 0; return wide_approximate_model(h, u, x, q, qv, Q, r, R, P, A, B, W); }
    std::vector<double> tp;
    rat_rc rc = linearize_slot0(h, u, x, &tp, domain_fail);
    if (rc) return rc;
    const int n = h->n, m = h->m, N = h->N;
    for (int t = 0; t < N; ++t) {
        const double *ts = &tp[(size_t)t * TSTRIDE];
        if (q) q[t] = ts[TS_q];
        for (int i = 0; i < n; ++i) {
            if (qv) qv[(size_t)t * n + i] = ts[TS_QR + i];
            for (int jj = 0; jj < n; ++jj) {
                if (Q) Q[(size_t)t * n * n + i + n * jj] = ts[TS_CPOS(i, jj)];
                if (A) A[(size_t)t * n * n + i + n * jj] = ts[TS_ZPOS(i, jj)];
            }
            for (int g = 0; g < m; ++g) if (B) B[(size_t)t * n * m + i + n * g] = ts[TS_ZPOS(i, 12 + g)];
        }
        for (int g = 0; g < m; ++g) {
            if (r) r[(size_t)t * m + g] = ts[TS_QR + 12 + g];
            for (int jj = 0; jj < n; ++jj) if (P) P[(size_t)t * m * n + g + m * jj] = ts[TS_CPOS(12 + g, jj)];
            for (int g2 = 0; g2 < m; ++g2) if (R) R[(size_t)t * m * m + g + m * g2] = ts[TS_CPOS(12 + g, 12 + g2)];
        }
        if (W) memcpy(W + (size_t)t * n * n, h->hW.data() + (h->W_tv ? (size_t)t * n * n : 0), sizeof(double) * n * n);   // :312
    }
    const double *tt = &tp[(size_t)N * TSTRIDE];
    if (q) q[N] = tt[TT_q];
    for (int i = 0; i < n; ++i) {
        if (qv) qv[(size_t)N * n + i] = tt[TT_QV + i];
        for (int jj = 0; jj < n; ++jj) if (Q) Q[(size_t)N * n * n + i + n * jj] = tt[TT_Q + i * 12 + jj];
    }
    return RAT_OK;
}

// ApproximationResult arrays (reference layout) -> one padded tile bundle
static void pack_tiles(const rat_handle h, const double *q, const double *qv, const double *Q, const double *r, const double *R,
                       const double *P, const double *A, const double *B, std::vector<double> &tp) {
    const int n = h->n, m = h->m, N = h->N;
    tp.assign(h->st.tile_stride, 0.0);
    for (int t = 0; t < N; ++t) {
        double *ts = &tp[(size_t)t * TSTRIDE];
        ts[TS_q] = q[t];
        for (int i = 0; i < n; ++i) {
            ts[TS_QR + i] = qv[(size_t)t * n + i];
            for (int jj = 0; jj < n; ++jj) {
                ts[TS_CPOS(i, jj)] = Q[(size_t)t * n * n + i + n * jj];
                ts[TS_ZPOS(i, jj)] = A[(size_t)t * n * n + i + n * jj];
            }
            for (int g = 0; g < m; ++g) ts[TS_ZPOS(i, 12 + g)] = B[(size_t)t * n * m + i + n * g];
        }
        for (int g = 0; g < m; ++g) {
            ts[TS_QR + 12 + g] = r[(size_t)t * m + g];
            for (int jj = 0; jj < n; ++jj) ts[TS_CPOS(12 + g, jj)] = P[(size_t)t * m * n + g + m * jj];
            for (int g2 = 0; g2 < m; ++g2) ts[TS_CPOS(12 + g, 12 + g2)] = R[(size_t)t * m * m + g + m * g2];
        }
        for (int g = m; g < RAT_MP; ++g) ts[TS_CPOS(12 + g, 12 + g)] = 1.0;
    }
    double *tt = &tp[(size_t)N * TSTRIDE];
    tt[TT_q] = q[N];
    for (int i = 0; i < n; ++i) {
        tt[TT_QV + i] = qv[(size_t)N * n + i];
        for (int jj = 0; jj < n; ++jj) tt[TT_Q + i * 12 + jj] = Q[(size_t)N * n * n + i + n * jj];
    }
}

static rat_rc unpack_dump(rat_handle h, double *s, double *sv, double *S, double *g, double *G, double *H) {
    const int n = h->n, m = h->m, N = h->N;
    std::vector<double> dp((size_t)(N + 1) * DUMP_STRIDE);
    HIPCHK(hipMemcpy(dp.data(), h->d_dump, dp.size() * 8, hipMemcpyDeviceToHost));
    for (int t = 0; t <= N; ++t) {
        const double *d = &dp[(size_t)t * DUMP_STRIDE];
        if (s) s[t] = d[DUMP_s];
        for (int i = 0; i < n; ++i) {
            if (sv) sv[(size_t)t * n + i] = d[DUMP_SV + i];
            // Symmetric(S): upper triangle mirrored (ileqg.jl:391)
            for (int jj = 0; jj < n; ++jj) if (S) S[(size_t)t * n * n + i + n * jj] = (i <= jj) ? d[DUMP_S + i * 12 + jj] : d[DUMP_S + jj * 12 + i];
        }
        if (t == N) continue;
        for (int a = 0; a < m; ++a) {
            if (g) g[(size_t)t * m + a] = d[DUMP_g + a];
            for (int jj = 0; jj < n; ++jj) if (G) G[(size_t)t * m * n + a + m * jj] = d[DUMP_G + a * 12 + jj];
            for (int a2 = 0; a2 < m; ++a2) if (H) H[(size_t)t * m * m + a + m * a2] = d[DUMP_H + a * 4 + a2];
        }
    }
    return RAT_OK;
}

extern "C" rat_rc rat_dp_gain_sweep(rat_handle h, const double *q, const double *qv, const double *Q, const double *r,
                                    const double *R, const double *P, const double *A, const double *B, double theta,
                                    double *mu, double *delta, double *L, double *dl, int32_t *status,
                                    double *s, double *sv, double *S, double *g, double *G, double *H) {
    if (!h || !q || !qv || !Q || !r || !R || !P || !A || !B || !mu || !delta) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide)
        return wide_dp(h, true, 1, q, qv, Q, r, R, P, A, B, &theta, mu, delta, nullptr, nullptr, nullptr, L, dl, nullptr, status, s, sv, S, g, G, H);
    StateDev st;
    rat_rc rc = op_prepare(h, theta, *mu, *delta, &st);
    if (rc) return rc;
    std::vector<double> tp;
    pack_tiles(h, q, qv, Q, r, R, P, A, B, tp);
    HIPCHK(hipMemcpy(st.tiles, tp.data(), tp.size() * 8, hipMemcpyHostToDevice));
    SweepArgs a = sweep_args(h, st, 0);
    a.op_out = h->d_opout; a.dump = h->d_dump;
    launch_sweep(a, 1, true, true, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    double oo[2];
    HIPCHK(hipMemcpy(oo, h->d_opout, 16, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mu, st.mu, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(delta, st.delta, 8, hipMemcpyDeviceToHost));
    if (status) *status = (int32_t)oo[1];
    std::vector<double> Lp((size_t)h->N * LSTR), dlp((size_t)h->N * USTR);
    HIPCHK(hipMemcpy(Lp.data(), st.L, Lp.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dlp.data(), st.dl, dlp.size() * 8, hipMemcpyDeviceToHost));
    if (L) unpad_L(h, Lp, L);
    if (dl) unpad_u(h, dlp, dl);
    return unpack_dump(h, s, sv, S, g, G, H);
}

extern "C" rat_rc rat_dp_policy_eval(rat_handle h, const double *q, const double *qv, const double *Q, const double *r,
                                     const double *R, const double *P, const double *A, const double *B, const double *L,
                                     const double *dl, double theta, double mu, int32_t *status,
                                     double *s, double *sv, double *S, double *g, double *G, double *H) {
    if (!h || !q || !qv || !Q || !r || !R || !P || !A || !B || !L) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide)
        return wide_dp(h, false, 1, q, qv, Q, r, R, P, A, B, &theta, nullptr, nullptr, &mu, L, dl, nullptr, nullptr, nullptr, status, s, sv, S, g, G, H);
    StateDev st;
    rat_rc rc = op_prepare(h, theta, mu, h->opts.delta_0, &st);
    if (rc) return rc;
    std::vector<double> tp, Lp, dlp;
    pack_tiles(h, q, qv, Q, r, R, P, A, B, tp);
    HIPCHK(hipMemcpy(st.tiles, tp.data(), tp.size() * 8, hipMemcpyHostToDevice));
    pad_L(h, L, Lp);
    HIPCHK(hipMemcpy(st.L, Lp.data(), Lp.size() * 8, hipMemcpyHostToDevice));
    SweepArgs a = sweep_args(h, st, 3);
    if (dl) { pad_u(h, dl, dlp); HIPCHK(hipMemcpy(h->d_dlin, dlp.data(), dlp.size() * 8, hipMemcpyHostToDevice)); a.dl_in = h->d_dlin; }
    a.mu_op = mu; a.op_out = h->d_opout; a.dump = h->d_dump;
    launch_sweep(a, 1, false, true, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    double oo[2];
    HIPCHK(hipMemcpy(oo, h->d_opout, 16, hipMemcpyDeviceToHost));
    if (status) *status = (int32_t)oo[1];
    return unpack_dump(h, s, sv, S, g, G, H);
}

// ---- batched sweeps on caller-supplied tiles (generic closures: the host linearises, the device sweeps) -------------------------
// B ApproximationResults, sample slowest (q[B][N+1], qv[B][n(N+1)], ...): packed into the tile bundles of samples 0..B-1 and swept by
// the SAME kernels the solver uses (sweep_kernel<gain> / sweep_kernel<eval>), one wavefront per sample.
static rat_rc upload_tiles_batch(rat_handle h, StateDev &st, int64_t B, bool candidate, const double *q, const double *qv, const double *Q,
                                 const double *r, const double *R, const double *P, const double *A, const double *Bm) {
    const int n = h->n, m = h->m, N = h->N;
    const size_t ts = (size_t)st.tile_stride;
    std::vector<double> host((size_t)B * ts), tp;
    for (int64_t b = 0; b < B; ++b) {
        pack_tiles(h, q + b * (N + 1), qv + b * (size_t)n * (N + 1), Q + b * (size_t)n * n * (N + 1), r + b * (size_t)m * N,
                   R + b * (size_t)m * m * N, P + b * (size_t)m * n * N, A + b * (size_t)n * n * N, Bm + b * (size_t)n * m * N, tp);
        memcpy(&host[(size_t)b * ts], tp.data(), ts * 8);
    }
    if (st.tile_alias) {                                  // one bundle per sample, contiguous
        HIPCHK(hipMemcpyAsync(st.tiles, host.data(), host.size() * 8, hipMemcpyHostToDevice, h->stream));
    } else {                                              // bundle of slot (nominal 0 | candidate 0) of every sample
        for (int64_t b = 0; b < B; ++b) {
            const long slot = candidate ? cand_slot((int)b, 0, 0, st.E) : (long)b * (st.E + 1);
            HIPCHK(hipMemcpyAsync(st.tiles + (size_t)slot * ts, &host[(size_t)b * ts], ts * 8, hipMemcpyHostToDevice, h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));              // (host staging buffer goes out of scope)
    return RAT_OK;
}

static rat_rc batch_state(rat_handle h, int64_t B, const double *theta, const double *mu, const double *delta, StateDev *out) {
    if (!h->have_problem) return fail(RAT_ERR_NO_PROBLEM, "rat_problem_set was not called");
    if (h->wide) return fail(RAT_ERR_UNSUPPORTED, WIDE_OP_MSG);
    if (B < 1 || B > h->Bmax) return fail(RAT_ERR_ARG, "batch size exceeds max_batch of rat_create");
    HIPCHK(hipSetDevice(h->device));
    StateDev st = h->st; st.B = (int)B;
    HIPCHK(hipMemcpyAsync(h->d_theta, theta, B * 8, hipMemcpyHostToDevice, h->stream));
    launch_init_state(st, h->opd, h->d_theta, h->stream);            // status RUNNING, slot_nom 0, lsel 0, ls_active 0
    HIPCHK(hipMemcpyAsync(st.mu, mu, B * 8, hipMemcpyHostToDevice, h->stream));
    if (delta) HIPCHK(hipMemcpyAsync(st.delta, delta, B * 8, hipMemcpyHostToDevice, h->stream));
    *out = st;
    return RAT_OK;
}

extern "C" rat_rc rat_dp_gain_sweep_batch(rat_handle h, int64_t B, const double *q, const double *qv, const double *Q, const double *r,
                                          const double *R, const double *P, const double *A, const double *Bm, const double *theta,
                                          double *mu, double *delta, double *L, double *dl, int32_t *status) {
    if (!h || !q || !qv || !Q || !r || !R || !P || !A || !Bm || !theta || !mu || !delta) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide)
        return wide_dp(h, true, B, q, qv, Q, r, R, P, A, Bm, theta, mu, delta, nullptr, nullptr, nullptr, L, dl, nullptr, status,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    StateDev st;
    rat_rc rc = batch_state(h, B, theta, mu, delta, &st);
    if (rc) return rc;
    if ((rc = upload_tiles_batch(h, st, B, false, q, qv, Q, r, R, P, A, Bm))) return rc;
    prof_begin(h, RAT_K_SWEEP_GAIN, (long)B); launch_sweep_or_psweep(h, sweep_args(h, st, 0), (int)B, true); prof_end(h);   // solve_approximate_dp! per sample, mu restarts inside
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> sth((size_t)B);
    HIPCHK(hipMemcpy(mu, st.mu, B * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(delta, st.delta, B * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sth.data(), st.status, B * 4, hipMemcpyDeviceToHost));
    const int n = h->n, m = h->m, N = h->N;
    std::vector<double> Lp((size_t)B * N * LSTR), dlp((size_t)B * N * USTR), Lone((size_t)N * LSTR), dlone((size_t)N * USTR);
    HIPCHK(hipMemcpy(Lp.data(), st.L, Lp.size() * 8, hipMemcpyDeviceToHost));             // half 0 (lsel = 0 after init)
    HIPCHK(hipMemcpy(dlp.data(), st.dl, dlp.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b) {
        if (status) status[b] = (sth[(size_t)b] == ST_RUNNING) ? 0 : sth[(size_t)b];
        if (L) { Lone.assign(Lp.begin() + (size_t)b * N * LSTR, Lp.begin() + (size_t)(b + 1) * N * LSTR); unpad_L(h, Lone, L + (size_t)b * m * n * N); }
        if (dl) { dlone.assign(dlp.begin() + (size_t)b * N * USTR, dlp.begin() + (size_t)(b + 1) * N * USTR); unpad_u(h, dlone, dl + (size_t)b * m * N); }
    }
    return RAT_OK;
}

extern "C" rat_rc rat_dp_policy_eval_batch(rat_handle h, int64_t B, const double *q, const double *qv, const double *Q, const double *r,
                                           const double *R, const double *P, const double *A, const double *Bm, const double *L,
                                           const double *theta, const double *mu, double *value, int32_t *status) {
    if (!h || !q || !qv || !Q || !r || !R || !P || !A || !Bm || !L || !theta || !mu || !value) return fail(RAT_ERR_ARG, "null");
    if (h->have_problem && h->wide)
        return wide_dp(h, false, B, q, qv, Q, r, R, P, A, Bm, theta, nullptr, nullptr, mu, L, nullptr, nullptr, nullptr, value, status,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    StateDev st;
    rat_rc rc = batch_state(h, B, theta, mu, nullptr, &st);
    if (rc) return rc;
    st.E = 1;                                             // one candidate per sample (candidate 0), whatever the handle speculates
    if (h->st.E != 1 && !st.tile_alias) return fail(RAT_ERR_UNSUPPORTED, "batched tile sweeps need a handle created with spec_eps = 1");
    if ((rc = upload_tiles_batch(h, st, B, true, q, qv, Q, r, R, P, A, Bm))) return rc;
    const int n = h->n, m = h->m, N = h->N;
    std::vector<double> Lp((size_t)B * N * LSTR, 0.0), Lone;
    for (int64_t b = 0; b < B; ++b) {
        pad_L(h, L + (size_t)b * m * n * N, Lone);
        memcpy(&Lp[(size_t)b * N * LSTR], Lone.data(), Lone.size() * 8);
    }
    HIPCHK(hipMemcpyAsync(st.L, Lp.data(), Lp.size() * 8, hipMemcpyHostToDevice, h->stream));
    std::vector<int> ones((size_t)B, 1);
    HIPCHK(hipMemcpyAsync(st.ls_active, ones.data(), B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(st.flag_c, 0, B * 4, h->stream));
    prof_begin(h, RAT_K_SWEEP_EVAL, (long)B); launch_sweep_or_psweep(h, sweep_args(h, st, 1), (int)B, false); prof_end(h);   // solve_approximate_dp (dl = nothing) per sample
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> fl((size_t)B);
    HIPCHK(hipMemcpy(value, st.value_c, B * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fl.data(), st.flag_c, B * 4, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b) {
        if (status) status[b] = fl[(size_t)b] ? RAT_ST_M_NOT_PD_GAIN : 0;
        if (fl[(size_t)b]) value[b] = INFINITY;
    }
    return RAT_OK;
}

// ---- Cross-Entropy loop -----------------------------------------------------------------------------------
extern "C" void rat_ce_default(rat_ce_solver *c) {                       // :100-127
    memset(c, 0, sizeof(*c));
    c->num_samples = 10; c->num_elite = 3; c->iter_max = 5; c->lambda = 0.5; c->use_theta_max = 0;
    c->mu_init = 1.0; c->sigma_init = 2.0; c->mu = c->mu_init; c->sigma = c->sigma_init;
    c->theta_max = 0.0; c->theta_min = INFINITY; c->iter_current = 0;
}
extern "C" void rat_ce_initialize(rat_ce_solver *c) {                    // :133-138
    c->iter_current = 0; c->mu = c->mu_init; c->sigma = c->sigma_init; c->theta_max = 0.0; c->theta_min = INFINITY;
}
extern "C" rat_rc rat_ce_set_stream(rat_handle h, const double *z, int64_t nz) {
    if (!h) return RAT_ERR_ARG;
    h->z = z; h->nz = nz; h->zpos = 0; h->internal_rng = false;
    return RAT_OK;
}
static uint64_t splitmix64(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
extern "C" rat_rc rat_ce_seed(rat_handle h, uint64_t seed) {
    if (!h) return RAT_ERR_ARG;
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) h->rs[i] = splitmix64(x);
    h->internal_rng = true; h->have_spare = false; h->z = nullptr; h->nz = 0; h->zpos = 0;
    h->zfifo.clear(); h->zfifo_pos = 0;
    return RAT_OK;
}
extern "C" int64_t rat_ce_stream_pos(rat_handle h) { return h ? h->zpos : -1; }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static double next_uniform(rat_handle h) {                              // xoshiro256++, 53-bit mantissa in (0,1]
    uint64_t *s = h->rs;
    const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return ((double)(result >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}
static double raw_normal(rat_handle h) {                                // Box-Muller on the built-in generator (both outputs are used)
    if (h->have_spare) { h->have_spare = false; return h->spare; }
    const double u1 = next_uniform(h), u2 = next_uniform(h);
    const double rr = std::sqrt(-2.0 * std::log(u1)), ang = 6.283185307179586476925286766559 * u2;
    h->spare = rr * std::sin(ang); h->have_spare = true;
    return rr * std::cos(ang);
}
static bool next_normal(rat_handle h, double *z) {
    if (!h->internal_rng) {
        if (!h->z || h->zpos >= h->nz) return false;
        *z = h->z[h->zpos++];
        return true;
    }
    h->zpos++;
    if (h->zfifo_pos < h->zfifo.size()) { *z = h->zfifo[h->zfifo_pos++]; return true; }   // drawn ahead (prefill_normals): same sequence
    *z = raw_normal(h);
    return true;
}
// Draw normals of the built-in generator ahead of their use (called while a batch runs on the device): the sequence next_normal hands
// out is unchanged, only when the Box-Muller arithmetic happens.
static void prefill_normals(rat_handle h, int64_t count) {
    if (!h->internal_rng || count <= 0) return;
    if (h->zfifo_pos > 0) { h->zfifo.erase(h->zfifo.begin(), h->zfifo.begin() + (std::ptrdiff_t)h->zfifo_pos); h->zfifo_pos = 0; }
    while ((int64_t)h->zfifo.size() < count) h->zfifo.push_back(raw_normal(h));
}

extern "C" rat_rc rat_ce_get_positive_samples(rat_handle h, double mu, double sigma, int64_t num, double *theta) {   // :233-246
    if (!h || !theta) return fail(RAT_ERR_ARG, "null");
    int64_t k = 0, guard = 0;
    while (true) {
        double z;
        if (!next_normal(h, &z)) return fail(RAT_ERR_STREAM_DRY, "standard-normal stream exhausted");
        const double th = mu + sigma * z;                               // rand(rng, Normal(mu, sigma))
        if (th > 0.0) theta[k++] = th;
        if (k >= num) break;
        if (++guard > (int64_t)1 << 40) return fail(RAT_ERR_DIVERGED, "get_positive_samples did not terminate");
    }
    return RAT_OK;
}

extern "C" rat_rc rat_ce_compute_cost(rat_handle h, const double *x0, const double *u0, const double *theta, int64_t B,
                                      double kl_bound, double *cost) {   // :173-195
    rat_rc rc = rat_ileqg_solve_batch(h, x0, u0, theta, B, cost, nullptr, nullptr, nullptr);
    if (rc) return rc;
    for (int64_t i = 0; i < B; ++i) cost[i] = cost[i] + kl_bound / theta[i];   // :193
    return RAT_OK;
}

extern "C" rat_rc rat_ce_begin_step(rat_ce_solver *c) { if (!c) return RAT_ERR_ARG; c->iter_current += 1; return RAT_OK; }   // :259

extern "C" rat_rc rat_ce_draw(rat_handle h, const rat_ce_solver *c, double *theta) {   // :266-279
    if (!h || !c) return fail(RAT_ERR_ARG, "null");
    if (c->iter_current == 1) return rat_ce_get_positive_samples(h, c->mu_init, c->sigma_init, c->num_samples, theta);
    return rat_ce_get_positive_samples(h, c->mu, c->sigma, c->num_samples, theta);
}

extern "C" rat_rc rat_ce_draw_stream(const rat_ce_solver *c, const double *z, int64_t nz, int64_t *zpos, double *theta) {
    if (!c || !z || !zpos || !theta) return fail(RAT_ERR_ARG, "null");
    const double mu = (c->iter_current == 1) ? c->mu_init : c->mu, sigma = (c->iter_current == 1) ? c->sigma_init : c->sigma;
    int64_t k = 0;
    while (k < c->num_samples) {                                          // get_positive_samples :233-246
        if (*zpos >= nz) return fail(RAT_ERR_STREAM_DRY, "standard-normal stream exhausted");
        const double th = mu + sigma * z[(*zpos)++];
        if (th > 0.0) theta[k++] = th;
    }
    return RAT_OK;
}

extern "C" rat_rc rat_ce_update(rat_ce_solver *c, const double *theta, const double *cost, int32_t *redraw) {   // :291-334
    if (!c || !theta || !cost || !redraw) return fail(RAT_ERR_ARG, "null");
    const int64_t B = c->num_samples;
    if (c->num_elite < 1 || c->num_elite > B) return fail(RAT_ERR_ARG, "num_elite out of range");
    int64_t num_inf = 0;
    for (int64_t i = 0; i < B; ++i) num_inf += std::isinf(cost[i]) ? 1 : 0;       // :291
    const int64_t num_valid = B - num_inf;
    const double thresh = std::max((double)c->num_elite, (double)B * c->lambda);
    *redraw = 1;
    if (c->iter_current == 1 && (double)num_valid < thresh) {                      // :293-298
        c->mu_init *= c->lambda; c->sigma_init *= c->lambda;
        return RAT_OK;
    } else if (c->iter_current == 1 && num_valid == B) {                           // :299-305
        c->mu_init /= c->lambda; c->sigma_init /= c->lambda;
    } else if ((double)num_valid >= thresh) {                                      // :306
    } else {
        return RAT_OK;                                                              // redraw with unchanged parameters
    }
    *redraw = 0;
    for (int64_t i = 0; i < B; ++i) {                                              // :314-324 (if / elseif)
        if (std::isinf(cost[i])) continue;
        if (theta[i] < c->theta_min) c->theta_min = theta[i];
        else if (theta[i] > c->theta_max) c->theta_max = theta[i];
    }
    std::vector<int64_t> idx(B);
    for (int64_t i = 0; i < B; ++i) idx[i] = i;
    // sort(by = cost) is stable and only its first num_elite entries are read (:326-330): a partial sort under the total order
    // (isless(cost), original index) yields exactly those entries -- NaN last, ties in input order -- without ordering the other 90 %
    std::partial_sort(idx.begin(), idx.begin() + (std::ptrdiff_t)c->num_elite, idx.end(), [&](int64_t a, int64_t b) {
        const double x = cost[a], y = cost[b];
        const bool xn = x != x, yn = y != y;
        if (xn || yn) return xn ? (yn && a < b) : true;
        if (x < y) return true;
        if (y < x) return false;
        const bool xs = std::signbit(x), ys = std::signbit(y);                       // isless(-0.0, 0.0) is true
        if (xs != ys) return xs;
        return a < b;
    });
    double sum = 0;
    for (int64_t i = 0; i < c->num_elite; ++i) sum += theta[idx[i]];
    const double mu_new = sum / (double)c->num_elite;                              // :329
    double ss = 0;
    for (int64_t i = 0; i < c->num_elite; ++i) ss += (theta[idx[i]] - mu_new) * (theta[idx[i]] - mu_new);
    c->mu = mu_new; c->sigma = std::sqrt(ss / (double)c->num_elite);              // :330-334 (population std)
    return RAT_OK;
}

extern "C" rat_rc rat_ce_step(rat_handle h, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                              double *theta_out, double *cost_out) {              // :252-335
    if (!h || !c) return fail(RAT_ERR_ARG, "null");
    if (c->num_samples > h->Bmax) return fail(RAT_ERR_ARG, "num_samples exceeds max_batch of rat_create");
    std::vector<double> theta(c->num_samples), cost(c->num_samples);
    rat_ce_begin_step(c);
    for (int redraws = 0;; ++redraws) {
        if (redraws > 1000) return fail(RAT_ERR_DIVERGED, "CE redraw loop cut after 1000 redraws (reference would spin, App. B.11)");
        rat_rc rc = rat_ce_draw(h, c, theta.data());
        if (rc) return rc;
        h->prefill_want = c->num_samples + c->num_samples / 8;       // the next draw's normals (with room for rejected theta <= 0), drawn under this batch
        rc = rat_ce_compute_cost(h, x0, u0, theta.data(), c->num_samples, kl_bound, cost.data());
        h->prefill_want = 0;
        if (rc) return rc;
        c->n_solves += c->num_samples;
        if (redraws) c->n_redraws++;
        int32_t redraw = 0;
        if ((rc = rat_ce_update(c, theta.data(), cost.data(), &redraw))) return rc;
        if (!redraw) break;
    }
    if (theta_out) memcpy(theta_out, theta.data(), theta.size() * 8);
    if (cost_out) memcpy(cost_out, cost.data(), cost.size() * 8);
    return RAT_OK;
}

// ---- solve! with the Cross-Entropy loop resident on the device (ce_device.hip) -------------------------------------------------------
// One enqueue chain per solve!: [draw -> batch -> update] x iter_max, the final solve at theta_opt, the record back, ONE host wait.  The
// host only stays ahead of the device with standard normals: an injected stream is uploaded up front; the built-in generator (host
// xoshiro256++ / Box-Muller: its libm transcendentals stay on the host so the sequence is the host path's) fills a pinned buffer one
// slot ahead while the previous batch runs.  Rare events -- a redraw (:293-298, :306) consumed a slot, the uploaded normals ran out, the
// final solve failed (:410-413) -- are seen after the wait and handled by enqueuing what is left.
static bool ce_device_usable(rat_handle h, const rat_ce_solver *c) {
    return h->ce_device && h->have_problem && c->num_samples >= 1 && c->num_samples <= CE_DEV_MAX_B && c->num_samples <= h->Bmax &&
           c->num_elite >= 1 && c->num_elite <= c->num_samples && pick_path(h, (int)c->num_samples) != PATH_ROUNDS;
}
static rat_rc ce_ensure_buffers(rat_handle h, size_t need_z) {
    if (!h->d_ce) {
        HIPCHK(hipMalloc((void **)&h->d_ce, sizeof(CeDev)));
        HIPCHK(hipHostMalloc((void **)&h->h_ce, 2 * sizeof(CeDev), hipHostMallocDefault));       // [0] upload image, [1] read-back
        HIPCHK(hipMalloc((void **)&h->d_ce_theta, sizeof(double) * CE_DEV_MAX_B));
        HIPCHK(hipMalloc((void **)&h->d_ce_cost, sizeof(double) * CE_DEV_MAX_B));
        // (a draw that runs dry writes nothing: the batches enqueued behind it then solve whatever the buffer holds before the host sees
        //  CE_ERR_DRY and repeats the chain -- keep that a benign problem, theta = 1, not uninitialised bits)
        std::vector<double> ones(CE_DEV_MAX_B, 1.0);
        HIPCHK(hipMemcpy(h->d_ce_theta, ones.data(), sizeof(double) * CE_DEV_MAX_B, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(h->d_ce_cost, 0, sizeof(double) * CE_DEV_MAX_B));
    }
    if (need_z > h->cap_cez) {                       // (grown outside any chain: the caller has synchronised)
        // The standard normals live in pinned host memory that the draw kernel reads in place (a draw touches ~12 KB of it once): no copy
        // command sits between a batch and the bookkeeping launch behind it.
        const size_t cap = std::max<size_t>(need_z, 2 * h->cap_cez);
        double *hz = nullptr;
        HIPCHK(hipHostMalloc((void **)&hz, cap * 8, hipHostMallocDefault));
        if (h->h_cez) { memcpy(hz, h->h_cez, h->cap_cez * 8); (void)hipHostFree(h->h_cez); }
        h->h_cez = hz; h->cap_cez = cap;
        void *dz = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dz, hz, 0));
        h->d_cez = (double *)dz;
    }
    return RAT_OK;
}

// rat_ce_update on the device: the update kernel of the device-resident loop (ce_step_kernel, do_update only) applied to host-supplied
// thetas / costs -- the same arithmetic and the same elite order as rat_ce_update (tests compare the two on costs that hold NaN, +-Inf,
// ties and both signed zeros).  The solve counters are the caller's business, as with rat_ce_update.
extern "C" rat_rc rat_ce_update_dev(rat_handle h, rat_ce_solver *c, const double *theta, const double *cost, int32_t *redraw) {
    if (!h || !c || !theta || !cost || !redraw) return fail(RAT_ERR_ARG, "null");
    const int64_t B = c->num_samples;
    if (B < 1 || B > CE_DEV_MAX_B) return fail(RAT_ERR_UNSUPPORTED, "rat_ce_update_dev: 1 <= num_samples <= 1024");
    if (c->num_elite < 1 || c->num_elite > B) return fail(RAT_ERR_ARG, "num_elite must be in [1, num_samples]");
    HIPCHK(hipSetDevice(h->device));
    rat_rc rc = ce_ensure_buffers(h, 1);
    if (rc) return rc;
    CeDev &up = h->h_ce[0];
    CeDev &back = h->h_ce[1];
    memset(&up, 0, sizeof(up));
    up.mu_init = c->mu_init; up.sigma_init = c->sigma_init; up.mu = c->mu; up.sigma = c->sigma; up.theta_max = c->theta_max; up.theta_min = c->theta_min;
    up.lambda = c->lambda; up.iter_current = c->iter_current; up.iter_max = c->iter_max; up.num_samples = B; up.num_elite = c->num_elite;
    up.use_theta_max = c->use_theta_max;
    HIPCHK(hipMemcpyAsync(h->d_ce, &up, sizeof(CeDev), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ce_theta, theta, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ce_cost, cost, sizeof(double) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    launch_ce_step(h->d_ce, h->d_cez, 0, h->d_ce_theta, h->d_ce_cost, 1, 0, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&back, h->d_ce, sizeof(CeDev), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    c->mu_init = back.mu_init; c->sigma_init = back.sigma_init; c->mu = back.mu; c->sigma = back.sigma;
    c->theta_max = back.theta_max; c->theta_min = back.theta_min;
    *redraw = back.redraw_pending;
    return RAT_OK;
}

static rat_rc ce_solve_device(rat_handle h, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                              double *theta_opt, double *x, double *l, double *L, double *value, double *theta_min, double *theta_max) {
    const int64_t B = c->num_samples;
    rat_rc rc = rat_set_initial(h, x0, u0);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    const size_t per_slot = (size_t)(3 * B + 128);              // normals provisioned per draw (a draw keeps P(theta > 0) of them: >= 1/2 unless sigma >> mu)
    const size_t slots_upfront = (size_t)std::min<int64_t>(std::max<int64_t>(c->iter_max, 0) + 2, 16);   // pinned up front; stage_to() grows it for longer chains
    // what the host holds of the stream: injected -> z[zpos..nz); built-in -> the FIFO of normals drawn ahead + the generator
    const bool internal = h->internal_rng;
    const size_t inj_avail = internal ? 0 : (size_t)std::max<int64_t>(0, (h->z ? h->nz - h->zpos : 0));
    if ((rc = ce_ensure_buffers(h, internal ? per_slot * slots_upfront : std::min(inj_avail, per_slot * slots_upfront) + 1))) return rc;
    size_t filled = 0, uploaded = 0;                            // h_cez[0..filled) generated / staged, [0..uploaded) on the device
    // Normals the built-in generator produced for this chain but the chain did not consume go back to the FIFO, in order, on EVERY way
    // out (an error return used to drop them: the next call then saw a different sequence than the host loop would have).
    struct GiveBack {
        rat_handle h; const bool on; size_t &filled; size_t consumed = 0;
        ~GiveBack() {
            if (!on || consumed >= filled) return;
            std::vector<double> rest(h->h_cez + consumed, h->h_cez + filled);
            rest.insert(rest.end(), h->zfifo.begin() + (std::ptrdiff_t)h->zfifo_pos, h->zfifo.end());
            h->zfifo.swap(rest); h->zfifo_pos = 0;
        }
    } give_back{h, internal, filled};
    auto stage_to = [&](size_t want) -> rat_rc {                // make h_cez hold `want` normals (or all the injected stream has)
        if (!internal) want = std::min(want, inj_avail);
        if (want > h->cap_cez) {
            HIPCHK(hipStreamSynchronize(h->stream));
            rat_rc r = ce_ensure_buffers(h, want);
            if (r) return r;
        }
        if (internal) {
            while (filled < want) {
                if (h->zfifo_pos < h->zfifo.size()) h->h_cez[filled++] = h->zfifo[h->zfifo_pos++];     // drawn ahead by an earlier call
                else h->h_cez[filled++] = raw_normal(h);
            }
        } else if (filled < want) {
            memcpy(h->h_cez + filled, h->z + h->zpos + filled, (want - filled) * 8);
            filled = want;
        }
        return RAT_OK;
    };
    auto upload = [&]() -> rat_rc {                             // (zero-copy: what is staged is visible to every launch enqueued after this point)
        uploaded = filled;
        return RAT_OK;
    };
    // the record
    CeDev &up = h->h_ce[0];
    CeDev &back = h->h_ce[1];
    memset(&up, 0, sizeof(up));
    up.mu_init = c->mu_init; up.sigma_init = c->sigma_init; up.mu = c->mu; up.sigma = c->sigma; up.theta_max = c->theta_max; up.theta_min = c->theta_min;
    up.lambda = c->lambda; up.iter_current = c->iter_current; up.iter_max = c->iter_max; up.num_samples = B; up.num_elite = c->num_elite;
    up.n_solves = c->n_solves; up.n_redraws = c->n_redraws; up.zpos = 0; up.use_theta_max = c->use_theta_max;
    up.theta_opt = c->use_theta_max ? c->theta_max : c->mu;     // (iter_max = 0: :375-382 on the initialised solver)
    HIPCHK(hipMemcpyAsync(h->d_ce, &up, sizeof(CeDev), hipMemcpyHostToDevice, h->stream));
    if ((rc = stage_to(per_slot))) return rc;
    if ((rc = upload())) return rc;
    BatchOut out; out.cost = h->d_ce_cost; out.kl_bound = kl_bound;
    int64_t slots = c->iter_max - c->iter_current, slots_done = 0, redraw_guard = 0;
    for (;;) {
        // draw_1 | batch_1 | update_1 + draw_2 | batch_2 | ... | update_n : one bookkeeping launch between two batches
        if (slots > 0) { prof_begin(h, RAT_K_CE, B); launch_ce_step(h->d_ce, h->d_cez, (long long)uploaded, h->d_ce_theta, h->d_ce_cost, 0, 1, h->stream); prof_end(h); }
        for (int64_t k = 0; k < slots; ++k) {
            if ((rc = run_batch(h, h->d_ce_theta, (int)B, out))) return rc;
            ++slots_done;
            const int more = (k + 1 < slots) ? 1 : 0;
            if (more) {            // the next draw's normals, generated / staged while this batch runs on the device
                if ((rc = stage_to(per_slot * (size_t)(slots_done + 1)))) return rc;
                if ((rc = upload())) return rc;
            }
            prof_begin(h, RAT_K_CE, B); launch_ce_step(h->d_ce, h->d_cez, (long long)uploaded, h->d_ce_theta, h->d_ce_cost, 1, more, h->stream); prof_end(h);
        }
        // the final solve at theta_opt (:390-414), speculatively behind the chain, and the record back with its outputs
        int32_t st = 0; double val = 0;
        rc = ileqg_solve_impl(h, x0, u0, 0.0, &h->d_ce->theta_opt, x, l, L, &val, &st, nullptr, nullptr, 0, nullptr, &back, h->d_ce, sizeof(CeDev));
        if (rc) return rc;
        if (back.error == CE_ERR_DRY) {
            // the draw needed more normals than were on the device: nothing was consumed.  Give it everything the host can (the built-in
            // generator: four more slots' worth; an injected stream: all of it) and repeat
            const size_t more = internal ? filled + 4 * per_slot : inj_avail;
            if (!internal && uploaded >= inj_avail) {
                h->zpos += (int64_t)inj_avail;
                return fail(RAT_ERR_STREAM_DRY, "standard-normal stream exhausted");
            }
            if ((rc = stage_to(more))) return rc;
            if ((rc = upload())) return rc;
            CeDev fix = back; fix.error = 0;                    // (draw_retry stays set: the repeated draw does not advance the iteration)
            h->h_ce[0] = fix;
            HIPCHK(hipMemcpyAsync(h->d_ce, &h->h_ce[0], sizeof(CeDev), hipMemcpyHostToDevice, h->stream));
            slots = (back.iter_max - back.iter_current) + 1;
            if (++redraw_guard > 1000) return fail(RAT_ERR_DIVERGED, "get_positive_samples did not terminate");
            continue;
        }
        if (back.iter_current < back.iter_max || back.redraw_pending) {      // redraws consumed slots: run what is left
            slots = (back.iter_max - back.iter_current) + (back.redraw_pending ? 1 : 0);
            if (++redraw_guard > 1000) return fail(RAT_ERR_DIVERGED, "CE redraw loop cut after 1000 redraws (reference would spin, App. B.11)");
            continue;
        }
        // the chain is complete: mirror the record, account for the consumed normals
        c->mu_init = back.mu_init; c->sigma_init = back.sigma_init; c->mu = back.mu; c->sigma = back.sigma;
        c->theta_max = back.theta_max; c->theta_min = back.theta_min; c->iter_current = back.iter_current;
        c->n_solves = back.n_solves; c->n_redraws = back.n_redraws;
        const size_t consumed = (size_t)back.zpos;
        h->zpos += (int64_t)consumed;
        give_back.consumed = consumed;                          // (the rest returns to the FIFO when this function leaves)
        double th_opt = back.theta_opt;
        const double tmin = c->theta_min, tmax = c->theta_max;
        for (int tries = 0;; ++tries) {                         // :390-414 (the first attempt ran behind the chain)
            if (tries > 10000) return fail(RAT_ERR_DIVERGED, "final-solve retry loop cut (reference would spin, App. B.15)");
            if (st == RAT_ST_OK || st == RAT_ST_ITER_MAX) {
                *theta_opt = th_opt;
                *value = val + kl_bound / th_opt;               // :406
                if (theta_min) *theta_min = tmin;
                if (theta_max) *theta_max = tmax;
                return RAT_OK;
            }
            th_opt = std::max(0.0, th_opt - c->sigma);          // :412
            c->n_final_retries++;
            if ((rc = rat_ileqg_solve(h, x0, u0, th_opt, x, l, L, &val, &st, nullptr, nullptr, 0, nullptr))) return rc;
        }
    }
}

extern "C" rat_rc rat_ce_solve(rat_handle h, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                               double *theta_opt, double *x, double *l, double *L, double *value,
                               double *theta_min, double *theta_max) {             // :364-415
    if (!h || !c || !theta_opt || !value) return fail(RAT_ERR_ARG, "null");
    if (!(kl_bound >= 0)) return fail(RAT_ERR_ARG, "KL Divergence Bound must be non-negative (:368)");
    rat_ce_initialize(c);                                                           // :369
    c->n_final_retries = 0;
    if (kl_bound > 0 && c->iter_max >= 0 && ce_device_usable(h, c) && (h->internal_rng || h->z))
        return ce_solve_device(h, c, x0, u0, kl_bound, theta_opt, x, l, L, value, theta_min, theta_max);
    double th_opt, tmin = 0.0, tmax = 0.0;
    if (kl_bound > 0) {
        while (c->iter_current < c->iter_max) {                                     // :371-373
            rat_rc rc = rat_ce_step(h, c, x0, u0, kl_bound, nullptr, nullptr);
            if (rc) return rc;
        }
        tmin = c->theta_min; tmax = c->theta_max;                                   // :374
        th_opt = c->use_theta_max ? tmax : c->mu;                                   // :375-382
    } else {
        th_opt = 0.0;                                                               // :388
    }
    for (int tries = 0;; ++tries) {                                                 // :390-414
        if (tries > 10000) return fail(RAT_ERR_DIVERGED, "final-solve retry loop cut (reference would spin, App. B.15)");
        int32_t st = 0; double val = 0;
        rat_rc rc = rat_ileqg_solve(h, x0, u0, th_opt, x, l, L, &val, &st, nullptr, nullptr, 0, nullptr);
        if (rc) return rc;
        if (st == RAT_ST_OK || st == RAT_ST_ITER_MAX) {
            *theta_opt = th_opt;
            if (kl_bound > 0) { *value = val + kl_bound / th_opt; if (theta_min) *theta_min = tmin; if (theta_max) *theta_max = tmax; }   // :406
            else { *value = val; if (theta_min) *theta_min = 0.0; if (theta_max) *theta_max = 0.0; }                                    // :408
            return RAT_OK;
        }
        th_opt = std::max(0.0, th_opt - c->sigma);                                   // :412
        c->n_final_retries++;
    }
}

// ---- RAT iLQR++ (Nelder-Mead over theta) ----------------------------------------------------------------------
extern "C" void rat_nm_default(rat_nm_solver *s) {                             // nelder_mead_bilevel_optimization.jl:102-128
    memset(s, 0, sizeof(*s));
    s->alpha = 1.0; s->beta = 2.0; s->gamma = 0.5; s->eps = 1e-2; s->lambda = 0.5; s->iter_max = 100;
    s->theta_high_init = 3.0; s->theta_low_init = 1e-8;
    s->theta_high = s->theta_high_init; s->theta_low = s->theta_low_init;
}
extern "C" void rat_nm_initialize(rat_nm_solver *s) {                          // :164-168 (c_high / c_low are left alone)
    s->iter_current = 0; s->theta_low = s->theta_low_init; s->theta_high = s->theta_high_init;
}
extern "C" rat_rc rat_nm_compute_cost(rat_handle h, const double *x0, const double *u0, double theta, double kl_bound, double *cost) {   // :134-158
    if (!h || !cost) return fail(RAT_ERR_ARG, "null");
    double v = 0;
    rat_rc rc = rat_ileqg_solve_batch(h, x0, u0, &theta, 1, &v, nullptr, nullptr, nullptr);
    if (rc) return rc;
    *cost = v + kl_bound / theta;                                              // Inf stays Inf
    return RAT_OK;
}

// ---- speculation ---------------------------------------------------------------------------------------------------------------
// One Nelder-Mead iteration is a chain of up to three dependent evaluations, a solve! each, and a batch of up to two workgroup
// generations takes one solve's time (solve_block_kernel: 0.28 ms for 1 ... 512 samples).  So every theta the sequential code CAN ask for
// is evaluated ahead of time, with the sequential code's own expressions (bit-equal thetas), and the code then runs unchanged against a
// table of (theta, cost): the six vertices of this iteration (round 2), the six of each of the twelve states the iteration can end in
// (round 4: two iterations per device call), and -- in rat_nm_solve -- both initial vertices with the first THREE iterations (nm_tree:
// 6, 78, 942 thetas for one, two, three iterations; bit-equal repeats are dropped).  The vertices that can become theta_low ride along,
// so the final solve (:346) is read out of the last batch's device state instead of being run again.  What the sequential code
// evaluates, in what order, and every number it produces are unchanged (n_solves counts its evaluations; n_batches the device calls:
// 1 instead of 6 for a solve of three iterations).
static bool same_bits(double a, double b) { return memcmp(&a, &b, 8) == 0; }
static bool nm_lookup(rat_handle h, double th, double *c) {
    for (size_t i = 0; i < h->nm_th.size(); ++i) if (same_bits(h->nm_th[i], th)) { *c = h->nm_c[i]; return true; }
    return false;
}
// the thetas one iteration from (theta_low = th_m, theta_high = th_h) can evaluate, as rat_nm_step_impl writes them
static void nm_vertices(const rat_nm_solver *s, double th_m, double th_h, double *th) {
    const double lo = s->theta_low_init;
    const double th_r = std::max(lo, th_m + s->alpha * (th_m - th_h));                   // reflection      :195-196
    th[0] = th_r;
    th[1] = std::max(lo, th_m + s->beta * (th_r - th_m));                                // expansion       :204-205
    th[2] = std::max(lo, th_m + s->gamma * (th_h - th_m));                               // contraction with theta_high kept  :232-233
    th[3] = std::max(lo, th_m + s->gamma * (th_r - th_m));                               // contraction after theta_high <- theta_r
    th[4] = (th_h + th_m) / 2;                                                           // shrink          :239
    th[5] = (th_r + th_m) / 2;
}
// evaluates the thetas of `list` not yet in the table, in ONE batch (or several when the handle is smaller)
// `carry`: thetas that ride along even when their cost is known -- the current vertices, one of which may be the theta_low the solve ends
// with: the final solve is read out of the LAST batch's device state, so they have to be in it (two more samples cost nothing)
static rat_rc nm_prefetch(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound, const std::vector<double> &list,
                          const std::vector<double> &carry = std::vector<double>()) {
    std::vector<double> todo;
    auto add = [&](double th) {
        for (double t2 : todo) if (same_bits(t2, th)) return;
        todo.push_back(th);
    };
    for (double th : list) {
        double c;
        if (!nm_lookup(h, th, &c)) add(th);
    }
    if (todo.empty()) return RAT_OK;
    for (double th : carry) if ((int64_t)todo.size() < h->Bmax) add(th);
    for (size_t o = 0; o < todo.size(); o += (size_t)h->Bmax) {
        const size_t nb = std::min(todo.size() - o, (size_t)h->Bmax);
        std::vector<double> c(nb);
        std::vector<int32_t> st(nb);
        rat_rc rc = rat_ileqg_solve_batch(h, x0, u0, todo.data() + o, (int64_t)nb, c.data(), st.data(), nullptr, nullptr);
        if (rc) return rc;
        s->n_batches += 1;
        for (size_t i = 0; i < nb; ++i) { h->nm_th.push_back(todo[o + i]); h->nm_c.push_back(c[i] + kl_bound / todo[o + i]); }
        h->nm_last.assign(todo.begin() + (std::ptrdiff_t)o, todo.begin() + (std::ptrdiff_t)(o + nb));
        h->nm_last_v.swap(c); h->nm_last_st.swap(st);
    }
    return RAT_OK;
}
static rat_rc nm_cost(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound, double th, double *c) {
    s->n_solves += 1;
    if (nm_lookup(h, th, c)) return RAT_OK;
    rat_rc rc = nm_prefetch(h, s, x0, u0, kl_bound, std::vector<double>(1, th));
    if (rc) return rc;
    return nm_lookup(h, th, c) ? RAT_OK : fail(RAT_ERR_HIP, "Nelder-Mead: evaluated theta missing from the table");
}
// every theta the next `depth` iterations from (theta_low = th_m, theta_high = th_h) can evaluate: 6, 78, 942 for depth 1, 2, 3
static void nm_tree(const rat_nm_solver *s, double th_m, double th_h, int depth, std::vector<double> &list) {
    double v[6];
    nm_vertices(s, th_m, th_h, v);
    list.insert(list.end(), v, v + 6);
    if (depth < 2) return;
    for (int k = 0; k < 6; ++k) {                                 // theta_high <- v[k]; the next iteration may swap the two (:184-187)
        nm_tree(s, th_m, v[k], depth - 1, list);
        nm_tree(s, v[k], th_m, depth - 1, list);
    }
}
// what to evaluate ahead of an iteration from (th_m, th_h): its vertices and the next iteration's
static void nm_plan(rat_handle h, const rat_nm_solver *s, double th_m, double th_h, std::vector<double> &list) {
    nm_tree(s, th_m, th_h, h->nm_depth >= 2 ? 2 : 1, list);
}
static size_t nm_unique(std::vector<double> &list) {               // drops bit-equal repeats, keeps the order
    std::vector<double> u;
    for (double th : list) {
        bool dup = false;
        for (double t2 : u) if (same_bits(t2, th)) { dup = true; break; }
        if (!dup) u.push_back(th);
    }
    list.swap(u);
    return list.size();
}

static rat_rc nm_step_impl(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound) {   // :174-252
    s->iter_current += 1;
    if (s->c_high < s->c_low) {                                                // :184-187
        std::swap(s->theta_low, s->theta_high);
        std::swap(s->c_low, s->c_high);
    }
    const double th_m = s->theta_low;
    double th[6];
    nm_vertices(s, th_m, s->theta_high, th);
    const double th_r = th[0], th_e = th[1], th_c_old = th[2], th_c_ref = th[3], th_s_old = th[4], th_s_ref = th[5];
    rat_rc rc;
    if (h->Bmax >= 6) {                                                         // (a smaller handle evaluates on demand)
        bool all = true;
        for (int i = 0; i < 6; ++i) { double c; all = all && nm_lookup(h, th[i], &c); }
        if (!all) {
            std::vector<double> list, carry;
            nm_plan(h, s, th_m, s->theta_high, list);
            if ((int64_t)nm_unique(list) + 2 > h->Bmax) {                        // too many for this handle: this iteration's six vertices only
                list.assign(th, th + 6);                                         // (built explicitly: after de-duplication the list's first six
                nm_unique(list);                                                 //  entries need not be them, and it may hold fewer than six)
            }
            if (h->nm_depth >= 1) carry.assign({th_m, s->theta_high});
            if ((rc = nm_prefetch(h, s, x0, u0, kl_bound, list, carry))) return rc;
        }
    }
    double c_r, c_e, c_c;
    if ((rc = nm_cost(h, s, x0, u0, kl_bound, th_r, &c_r))) return rc;
    if (c_r < s->c_low) {                                                       // :202
        if ((rc = nm_cost(h, s, x0, u0, kl_bound, th_e, &c_e))) return rc;
        if (c_e < c_r) { s->theta_high = th_e; s->c_high = c_e; }              // :209-220
        else { s->theta_high = th_r; s->c_high = c_r; }
    } else {
        bool took_r = false;
        if (c_r < s->c_high) { s->theta_high = th_r; s->c_high = c_r; took_r = true; }   // :227-230
        if ((rc = nm_cost(h, s, x0, u0, kl_bound, took_r ? th_c_ref : th_c_old, &c_c))) return rc;   // :232-234
        if (c_c > s->c_high) {                                                  // :238-240
            s->theta_high = took_r ? th_s_ref : th_s_old;
            if ((rc = nm_cost(h, s, x0, u0, kl_bound, s->theta_high, &s->c_high))) return rc;
        } else { s->theta_high = took_r ? th_c_ref : th_c_old; s->c_high = c_c; }   // :245-250
    }
    return RAT_OK;
}

// the table of evaluated thetas is valid for one (problem, iLEQG options / execution switches, x0, u0, kl_bound); the Nelder-Mead constants that
// shape the vertices do not matter: thetas are keys
static uint64_t nm_table_key(rat_handle h, const double *x0, const double *u0, double kl_bound) {
    uint64_t k = (1469598103934665603ull ^ h->problem_serial) * 1099511628211ull ^ (h->opts_serial << 32);
    auto mix = [&](const double *p, size_t cnt) {
        for (size_t i = 0; i < cnt; ++i) { uint64_t b; memcpy(&b, p + i, 8); k = (k ^ b) * 1099511628211ull; k ^= k >> 29; }
    };
    if (x0) mix(x0, (size_t)h->n);
    if (u0) mix(u0, (size_t)h->N * h->m);
    mix(&kl_bound, 1);
    return k | 1ull;
}
extern "C" rat_rc rat_nm_step(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound) {   // :174-252
    if (!h || !s) return fail(RAT_ERR_ARG, "null");
    // a caller's own loop over step! keeps what earlier steps evaluated ahead, as long as it is the same (problem, x0, u0, kl_bound)
    const uint64_t key = nm_table_key(h, x0, u0, kl_bound);
    if (key != h->nm_key) { h->nm_th.clear(); h->nm_c.clear(); h->nm_last.clear(); h->nm_last_v.clear(); h->nm_last_st.clear(); h->nm_key = key; }
    return nm_step_impl(h, s, x0, u0, kl_bound);
}

// general sizes: the trajectory, controls and gains of sample b of the batch that ran last (wide_solve_kernel keeps every sample's two
// (x, u) slots, its gains and which slot holds the result); value and status are the batch's own outputs
static rat_rc fetch_batch_sample_wide(rat_handle h, int b, double *x, double *l, double *L) {
    HIPCHK(hipSetDevice(h->device));
    const size_t xs = (size_t)(h->N + 1) * h->n, us = (size_t)h->N * h->m, Ls = us * h->n;
    int32_t *p_i = reinterpret_cast<int32_t *>(h->h_io);
    HIPCHK(hipMemcpyAsync(p_i, h->w_nom + b, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int nom = p_i[0];
    if (nom < 0 || nom > 1) return fail(RAT_ERR_HIP, "general-size solve: slot index out of range");
    if (x) HIPCHK(hipMemcpyAsync(x, h->w_xs + ((size_t)b * 2 + nom) * xs, xs * 8, hipMemcpyDeviceToHost, h->stream));
    if (l) HIPCHK(hipMemcpyAsync(l, h->w_us + ((size_t)b * 2 + nom) * us, us * 8, hipMemcpyDeviceToHost, h->stream));
    if (L) HIPCHK(hipMemcpyAsync(L, h->w_L + (size_t)b * Ls, Ls * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RAT_OK;
}

// x, l, L, value, status of sample b of the batch that ran last, as rat_ileqg_solve returns them (tile-sized problems)
static rat_rc fetch_batch_sample(rat_handle h, int b, double *x, double *l, double *L, double *value, int32_t *status) {
    HIPCHK(hipSetDevice(h->device));
    const size_t xs = (size_t)h->st.x_stride, us = (size_t)h->st.u_stride, Ls = (size_t)h->N * LSTR;
    const size_t need = xs + us + Ls;
    if (need > h->cap_sol) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->h_sol) (void)hipHostFree(h->h_sol);
        h->h_sol = nullptr; h->cap_sol = 0;
        HIPCHK(hipHostMalloc((void **)&h->h_sol, need * 8, hipHostMallocDefault));
        h->cap_sol = need;
    }
    double *const s_x = h->h_sol, *const s_u = s_x + xs, *const s_L = s_u + us;
    double *const s_s = reinterpret_cast<double *>(h->h_io) + 2;
    launch_pack_solution(h->st, b, x ? s_x : nullptr, l ? s_u : nullptr, L ? s_L : nullptr, s_s, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    const int st_h = (int)s_s[1];
    if (status) *status = st_h;
    if (value) *value = (st_h == 0 || st_h == 3) ? s_s[0] : INFINITY;
    if (x) { std::vector<double> xp(s_x, s_x + xs); unpad_x(h, xp, x); }
    if (l) { std::vector<double> up(s_u, s_u + us); unpad_u(h, up, l); }
    if (L) { std::vector<double> Lp(s_L, s_L + Ls); unpad_L(h, Lp, L); }
    return RAT_OK;
}

extern "C" rat_rc rat_nm_solve(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound,
                               double *theta_opt, double *x, double *l, double *L, double *value, int32_t *status) {   // :276-352
    if (!h || !s || !theta_opt || !value) return fail(RAT_ERR_ARG, "null");
    if (!(kl_bound >= 0)) return fail(RAT_ERR_ARG, "KL Divergence Bound must be non-negative (:279)");
    rat_nm_initialize(s);
    h->nm_th.clear(); h->nm_c.clear(); h->nm_last.clear(); h->nm_last_v.clear(); h->nm_last_st.clear();
    h->nm_key = nm_table_key(h, x0, u0, kl_bound);
    double th_opt;
    rat_rc rc;
    if (kl_bound > 0) {
        if (!s->has_c_high && !s->has_c_low && h->nm_depth >= 1 && h->Bmax >= 14) {
            // both initial vertices and the first iterations under either ordering, in one device call.  The ordering in which the vertex
            // at theta_low_init turns out the worse one (its kl_bound / theta term is what makes it so: the swap of :184-187) is followed
            // one level deeper when the handle is large enough -- 2 + 942 + 78 thetas: one batch on the one-wave-per-sample kernel covers
            // the initial pair and THREE iterations (0.39 ms instead of two batches of 0.28 ms).  A wrong guess costs nothing but a miss.
            std::vector<double> list;
            for (int deep = std::min(h->nm_depth, 3); deep >= 1; --deep) {
                list.assign({s->theta_high, s->theta_low});
                nm_tree(s, s->theta_high, s->theta_low, deep, list);
                nm_tree(s, s->theta_low, s->theta_high, std::min(deep, 2), list);
                if ((int64_t)nm_unique(list) <= h->Bmax) break;
            }
            if ((int64_t)list.size() > h->Bmax) list.resize((size_t)h->Bmax);
            if ((rc = nm_prefetch(h, s, x0, u0, kl_bound, list))) return rc;
        }
        if (!s->has_c_high) {                                                   // :283-293
            for (int guard = 0;; ++guard) {
                if (guard > 2000) return fail(RAT_ERR_DIVERGED, "theta_high halving loop cut");
                if ((rc = nm_cost(h, s, x0, u0, kl_bound, s->theta_high, &s->c_high))) return rc;
                s->has_c_high = 1;
                if (!std::isinf(s->c_high)) break;
                s->theta_high *= s->lambda; s->theta_high_init *= s->lambda;
            }
        }
        if (!s->has_c_low) {                                                    // :294-304
            for (int guard = 0;; ++guard) {
                if (guard > 2000) return fail(RAT_ERR_DIVERGED, "theta_low halving loop cut");
                if ((rc = nm_cost(h, s, x0, u0, kl_bound, s->theta_low, &s->c_low))) return rc;
                s->has_c_low = 1;
                if (!std::isinf(s->c_low)) break;
                s->theta_low *= s->lambda; s->theta_low_init *= s->lambda;
            }
        }
        for (;;) {                                                              // :306-324
            if ((rc = nm_step_impl(h, s, x0, u0, kl_bound))) return rc;
            const double c_mean = (s->c_low + s->c_high) / 2;
            const double stdev = std::sqrt(0.5 * ((s->c_high - c_mean) * (s->c_high - c_mean) + (s->c_low - c_mean) * (s->c_low - c_mean)));
            if (stdev < s->eps) break;
            if (s->iter_current == s->iter_max) break;
        }
        th_opt = s->theta_low;                                                  // :325
    } else {
        th_opt = 0.0;                                                           // :332
    }
    int32_t st = 0; double val = 0;
    int at = -1;                                                                // theta_opt among the samples of the batch that ran last?
    for (size_t i = 0; i < h->nm_last.size(); ++i) if (same_bits(h->nm_last[i], th_opt)) { at = (int)i; break; }
    if (at >= 0 && h->wide) {
        if ((rc = fetch_batch_sample_wide(h, at, x, l, L))) return rc;
        st = h->nm_last_st[(size_t)at]; val = h->nm_last_v[(size_t)at];
    } else if (at >= 0) { if ((rc = fetch_batch_sample(h, at, x, l, L, &val, &st))) return rc; }
    else if ((rc = rat_ileqg_solve(h, x0, u0, th_opt, x, l, L, &val, &st, nullptr, nullptr, 0, nullptr))) return rc;   // :346 (not in a try)
    if (status) *status = st;
    *theta_opt = th_opt;
    *value = (kl_bound > 0) ? val + kl_bound / th_opt : val;                    // :347-351
    return RAT_OK;
}

// ---- PETS (pets.jl) on the generative family ---------------------------------------------------------------------
static bool host_chol_lower(int n, const double *A, double *Lo) {          // column-major in/out
    std::fill(Lo, Lo + n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[j + n * j];
        for (int k = 0; k < j; ++k) d -= Lo[j + n * k] * Lo[j + n * k];
        if (!(d > 0.0)) return false;
        Lo[j + n * j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double v = A[i + n * j];
            for (int k = 0; k < j; ++k) v -= Lo[i + n * k] * Lo[j + n * k];
            Lo[i + n * j] = v / Lo[j + n * j];
        }
    }
    return true;
}

extern "C" rat_rc rat_pets_problem_set(rat_handle h, const rat_gen_problem_desc *d) {
    if (!h || !d) return fail(RAT_ERR_ARG, "null");
    HIPCHK(hipSetDevice(h->device));
    const rat_problem_desc &q = d->lq;
    const int n = q.n, m = q.m, N = q.N;
    if (q.model != RAT_MODEL_LQ) return fail(RAT_ERR_UNSUPPORTED, "generative family is built on RAT_MODEL_LQ");
    if (n < 1 || m < 1 || N < 1) return fail(RAT_ERR_ARG, "n, m, N must be positive");
    if (n > RAT_NP || m > RAT_MP) return fail(RAT_ERR_UNSUPPORTED, "kernels are compiled for n <= 12, m <= 4");
    if (!q.A || !q.B || !q.Q || !q.R || !q.P || !q.qv || !q.rv || !q.q0 || !q.Qf || !q.qvf) return fail(RAT_ERR_ARG, "a table pointer is null");
    if (d->noise_kind == 0 && (!d->nmean || !d->nchol)) return fail(RAT_ERR_ARG, "Gaussian noise needs nmean / nchol");
    if (d->tw2 > 0 && (!d->tmean2 || !d->tchol2)) return fail(RAT_ERR_ARG, "mixture noise needs tmean2 / tchol2");
    if (d->tw2 > 0 && d->noise_kind != 0)
        return fail(RAT_ERR_UNSUPPORTED, "the true-model mixture is defined over Gaussian model noise (one N(0,1) stream feeds both components)");
    HIPCHK(hipStreamSynchronize(h->stream));
    free_list(h->gen_allocs);
    GenDev g;
    memset(&g, 0, sizeof(g));
    g.n = n; g.m = m; g.N = N; g.cost_tv = q.cost_tv ? 1 : 0; g.noise_kind = d->noise_kind;
    g.q0f = q.q0f; g.kappa = q.kappa; g.l1u = d->l1u; g.nlo = d->nlo; g.nhi = d->nhi; g.tw2 = d->tw2;
    const int Nc = g.cost_tv ? N : 1;
    std::vector<double> Zt(192, 0.0), Ctab((size_t)Nc * 256, 0.0), lin((size_t)Nc * 16, 0.0), q0(Nc, 0.0), Qf(144, 0.0), qvf(16, 0.0);
    for (int i = 0; i < n; ++i) {
        for (int jj = 0; jj < n; ++jj) Zt[i * 16 + jj] = q.A[i + n * jj];
        for (int a = 0; a < m; ++a) Zt[i * 16 + 12 + a] = q.B[i + n * a];
    }
    for (int k = 0; k < Nc; ++k) {
        double *C = &Ctab[(size_t)k * 256];
        const double *Q = q.Q + (size_t)k * n * n, *R = q.R + (size_t)k * m * m, *P = q.P + (size_t)k * m * n;
        // c(k, x, u) is evaluated as written: 1/2 x'Qx + 1/2 u'Ru + u'Px; the symmetric part is what a quadratic form sees
        for (int i = 0; i < n; ++i) for (int jj = 0; jj < n; ++jj) C[i * 16 + jj] = 0.5 * (Q[i + n * jj] + Q[jj + n * i]);
        for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) C[(12 + a) * 16 + 12 + b] = 0.5 * (R[a + m * b] + R[b + m * a]);
        for (int a = 0; a < m; ++a) for (int jj = 0; jj < n; ++jj) { C[(12 + a) * 16 + jj] = P[a + m * jj]; C[jj * 16 + 12 + a] = P[a + m * jj]; }
        for (int i = 0; i < n; ++i) lin[(size_t)k * 16 + i] = q.qv[(size_t)k * n + i];
        for (int a = 0; a < m; ++a) lin[(size_t)k * 16 + 12 + a] = q.rv[(size_t)k * m + a];
        q0[k] = q.q0[k];
    }
    for (int i = 0; i < n; ++i) {
        for (int jj = 0; jj < n; ++jj) Qf[i * 12 + jj] = 0.5 * (q.Qf[i + n * jj] + q.Qf[jj + n * i]);
        qvf[i] = q.qvf[i];
    }
    auto pack_vec = [&](const double *v) { std::vector<double> o(16, 0.0); if (v) for (int i = 0; i < n; ++i) o[i] = v[i]; return o; };
    auto pack_low = [&](const double *Lm) { std::vector<double> o(192, 0.0); if (Lm) for (int i = 0; i < n; ++i) for (int jj = 0; jj <= i; ++jj) o[i * 16 + jj] = Lm[i + n * jj]; return o; };
    rat_rc rc;
#define UPG(field, vec) if ((rc = dev_upload(h, h->gen_allocs, &g.field, vec))) return rc
    UPG(Zt, Zt); UPG(Ctab, Ctab); UPG(lin, lin); UPG(q0, q0); UPG(Qf, Qf); UPG(qvf, qvf);
    { auto v = pack_vec(d->nmean); UPG(nmean, v); }
    { auto v = pack_low(d->nchol); UPG(nchol, v); }
    if (d->tw2 > 0) { auto v = pack_vec(d->tmean2); UPG(tmean2, v); auto w = pack_low(d->tchol2); UPG(tchol2, w); }
#undef UPG
    h->gen = g; h->gn = n; h->gm = m; h->gN = N; h->have_gen = true;
    return RAT_OK;
}

extern "C" void rat_pets_initialize(rat_pets_solver *s) {                     // pets.jl:70-74
    s->iter_current = 0;
    memcpy(s->mu, s->mu_init, sizeof(double) * s->N * s->m);
    memcpy(s->Sigma, s->Sigma_init, sizeof(double) * s->N * s->m * s->m);
}

static rat_rc grow(double **p, size_t *cap, size_t need) {
    if (need <= *cap) return RAT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(hipMalloc((void **)p, need * sizeof(double)));
    *cap = need;
    return RAT_OK;
}

// enqueue form (multi.cpp drives several devices from one thread): everything up to and including the copy of the costs into
// `cost` -- which should be pinned memory for the copy to be asynchronous -- is ordered on the handle's stream; no host wait.
// sample0 = global index of controls[0] among the whole batch's control samples.
rat_rc rat_pets_enqueue(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K, int32_t use_true_model, const double *zn,
                        const double *zu, uint64_t seed, int64_t sample0, double *cost);

// One compute_cost evaluation enqueued on the handle's stream.  x0 and the padded controls are packed into a pinned staging area and pulled
// into device memory by pets_stage_kernel; the costs either land in the handle's pinned h_pcost, written by pets_mean_kernel itself
// (cost == nullptr: the synchronous call below copies them out after its wait), or are copied to `cost` by the copy engine (the several-GPU
// driver's pinned landing zone, multi.cpp).  No copy-engine transfer sits between host and kernels on the synchronous path: each costs
// ~10 us of engine hand-over against a 20-50 us kernel at BASELINE config 5's 10k trajectories.
static rat_rc pets_enqueue_impl(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K, int32_t use_true_model,
                                const double *zn, const double *zu, uint64_t seed, int64_t sample0, double *cost) {
    if (!h || !x0 || !controls) return fail(RAT_ERR_ARG, "null");
    if (!h->have_gen) return fail(RAT_ERR_NO_PROBLEM, "rat_pets_problem_set was not called");
    if (S < 1 || K < 1) return fail(RAT_ERR_ARG, "S, K must be positive");
    HIPCHK(hipSetDevice(h->device));
    const int n = h->gn, m = h->gm, N = h->gN;
    const size_t ntraj = (size_t)S * K;
    rat_rc rc;
    const size_t nstage = XSTR + (size_t)S * N * USTR;
    HIPCHK(hipStreamSynchronize(h->stream));                  // (a previous enqueue may still be reading the staging area and the buffers)
    if ((rc = grow(&h->d_pin, &h->cap_pin, nstage))) return rc;
    if ((rc = grow(&h->d_ptraj, &h->cap_traj, ntraj))) return rc;
    if (nstage > h->cap_pstage) {
        if (h->h_pstage) (void)hipHostFree(h->h_pstage);
        h->h_pstage = nullptr; h->cap_pstage = 0;
        HIPCHK(hipHostMalloc((void **)&h->h_pstage, nstage * sizeof(double), hipHostMallocDefault));
        h->cap_pstage = nstage;
    }
    double *cost_dev = nullptr;
    if (cost) {
        if ((rc = grow(&h->d_pcost, &h->cap_cost, (size_t)S))) return rc;
        cost_dev = h->d_pcost;
    } else {
        if ((size_t)S > h->cap_hpcost) {
            if (h->h_pcost) (void)hipHostFree(h->h_pcost);
            h->h_pcost = nullptr; h->cap_hpcost = 0;
            HIPCHK(hipHostMalloc((void **)&h->h_pcost, (size_t)S * sizeof(double), hipHostMallocDefault));
            h->cap_hpcost = (size_t)S;
        }
        HIPCHK(hipHostGetDevicePointer((void **)&cost_dev, h->h_pcost, 0));
    }
    double *xp = h->h_pstage, *cp = h->h_pstage + XSTR;
    memset(xp, 0, XSTR * sizeof(double));
    for (int i = 0; i < n; ++i) xp[i] = x0[i];
    if (m == USTR) memcpy(cp, controls, (size_t)S * N * USTR * sizeof(double));
    else {
        memset(cp, 0, (size_t)S * N * USTR * sizeof(double));
        for (int64_t ii = 0; ii < S; ++ii) for (int t = 0; t < N; ++t) for (int a = 0; a < m; ++a)
            cp[((size_t)ii * N + t) * USTR + a] = controls[((size_t)ii * N + t) * m + a];
    }
    const double *stage_dev = nullptr;
    HIPCHK(hipHostGetDevicePointer((void **)&stage_dev, h->h_pstage, 0));
    launch_pets_stage(stage_dev, h->d_pin, (long)nstage, h->stream);
    PetsArgs a;
    a.g = h->gen; a.x0 = h->d_pin; a.controls = h->d_pin + XSTR; a.S = S; a.K = K; a.use_true = use_true_model ? 1 : 0;
    a.zn = nullptr; a.zu = nullptr; a.seed = seed; a.traj0 = (long)(sample0 * K); a.traj_cost = h->d_ptraj; a.cost = cost_dev;
    a.wave16 = h->pets_wave16;
    if (zn) {
        if ((rc = grow(&h->d_pzn, &h->cap_zn, ntraj * N * n))) return rc;
        HIPCHK(hipMemcpyAsync(h->d_pzn, zn, ntraj * N * n * 8, hipMemcpyHostToDevice, h->stream));
        a.zn = h->d_pzn;
        if (zu) {
            if ((rc = grow(&h->d_pzu, &h->cap_zu, ntraj * N))) return rc;
            HIPCHK(hipMemcpyAsync(h->d_pzu, zu, ntraj * N * 8, hipMemcpyHostToDevice, h->stream));
            a.zu = h->d_pzu;
        }
    }
    prof_begin(h, RAT_K_PETS, (int64_t)ntraj);
    launch_pets(a, h->stream);
    prof_end(h);
    HIPCHK(hipGetLastError());
    if (cost) HIPCHK(hipMemcpyAsync(cost, h->d_pcost, (size_t)S * 8, hipMemcpyDeviceToHost, h->stream));
    return RAT_OK;
}

extern "C" rat_rc rat_pets_compute_cost(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K,
                                        int32_t use_true_model, const double *zn, const double *zu, uint64_t seed, double *cost) {   // pets.jl:128-157
    if (!cost) return fail(RAT_ERR_ARG, "null");
    rat_rc rc = pets_enqueue_impl(h, x0, controls, S, K, use_true_model, zn, zu, seed, 0, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(cost, h->h_pcost, (size_t)S * sizeof(double));
    return RAT_OK;
}

rat_rc rat_pets_enqueue(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K, int32_t use_true_model, const double *zn,
                        const double *zu, uint64_t seed, int64_t sample0, double *cost) {
    if (!cost) return fail(RAT_ERR_ARG, "null");
    return pets_enqueue_impl(h, x0, controls, S, K, use_true_model, zn, zu, seed, sample0, cost);
}

extern "C" rat_rc rat_pets_sample_controls(const rat_pets_solver *s, const double *zc, double *controls) {   // pets.jl:206-216
    if (!s || !zc || !controls) return fail(RAT_ERR_ARG, "null");
    const int64_t S = s->num_control_samples, N = s->N, m = s->m;
    std::vector<double> Lc((size_t)N * m * m);
    for (int64_t t = 0; t < N; ++t)
        if (!host_chol_lower((int)m, s->Sigma + (size_t)t * m * m, &Lc[(size_t)t * m * m]))
            return fail(RAT_ERR_ARG, "Sigma_t is not positive definite (MvNormal would throw)");
    for (int64_t ii = 0; ii < S; ++ii)
        for (int64_t t = 0; t < N; ++t) {
            const double *z = zc + ((size_t)ii * N + t) * m, *Lt = &Lc[(size_t)t * m * m];
            for (int64_t a = 0; a < m; ++a) {
                double v = s->mu[t * m + a];
                for (int64_t b = 0; b <= a; ++b) v += Lt[a + m * b] * z[b];
                controls[((size_t)ii * N + t) * m + a] = v;                    // rand(rng, MvNormal(mu_t, Sigma_t))
            }
        }
    return RAT_OK;
}

extern "C" rat_rc rat_pets_update(rat_pets_solver *s, const double *controls, const double *cost, int64_t *elite_idx) {   // pets.jl:159-191
    if (!s || !controls || !cost) return fail(RAT_ERR_ARG, "null");
    const int64_t S = s->num_control_samples, E = s->num_elite, N = s->N, m = s->m;
    if (E < 2 || E > S) return fail(RAT_ERR_ARG, "num_elite must be in [2, num_control_samples]");
    std::vector<int64_t> idx(S);
    for (int64_t i = 0; i < S; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {      // sort(by = cost) :167, isless: NaN last
        const double x = cost[a], y = cost[b];
        if (x != x) return false;
        if (y != y) return true;
        return x < y || (x == y && std::signbit(x) && !std::signbit(y));        // isless(-0.0, 0.0) is true
    });
    for (int64_t e = 0; e < E; ++e) if (elite_idx) elite_idx[e] = idx[e];
    const double sf = s->smoothing_factor;
    for (int64_t t = 0; t < N; ++t)
        for (int64_t a = 0; a < m; ++a) {
            double mean = 0;
            for (int64_t e = 0; e < E; ++e) mean += controls[((size_t)idx[e] * N + t) * m + a];
            mean /= (double)E;                                                 // :183
            double var = 0;
            for (int64_t e = 0; e < E; ++e) { const double d = controls[((size_t)idx[e] * N + t) * m + a] - mean; var += d * d; }
            var /= (double)(E - 1);                                            // var = unbiased :184
            s->mu[t * m + a] = (1.0 - sf) * mean + sf * s->mu[t * m + a];       // :186
            for (int64_t b = 0; b < m; ++b) {                                  // Diagonal(var) :184, smoothing :187
                double *Sg = &s->Sigma[(size_t)t * m * m + a + m * b];
                *Sg = (1.0 - sf) * (a == b ? var : 0.0) + sf * *Sg;
            }
        }
    return RAT_OK;
}

extern "C" rat_rc rat_pets_step(rat_handle h, rat_pets_solver *s, const double *x0, int32_t use_true_model, const double *zc,
                                const double *zn, const double *zu, uint64_t seed, double *controls_out, double *cost_out) {   // pets.jl:193-245
    if (!h || !s || !zc) return fail(RAT_ERR_ARG, "null");
    if (!h->have_gen) return fail(RAT_ERR_NO_PROBLEM, "rat_pets_problem_set was not called");
    if (s->N != h->gN || s->m != h->gm) return fail(RAT_ERR_ARG, "solver N / m do not match the problem");
    const int64_t S = s->num_control_samples;
    s->iter_current += 1;
    std::vector<double> controls((size_t)S * s->N * s->m), cost(S);
    rat_rc rc = rat_pets_sample_controls(s, zc, controls.data());
    if (rc) return rc;
    if ((rc = rat_pets_compute_cost(h, x0, controls.data(), S, s->num_trajectory_samples, use_true_model, zn, zu, seed, cost.data()))) return rc;
    if ((rc = rat_pets_update(s, controls.data(), cost.data(), nullptr))) return rc;
    if (controls_out) memcpy(controls_out, controls.data(), controls.size() * 8);
    if (cost_out) memcpy(cost_out, cost.data(), cost.size() * 8);
    return RAT_OK;
}

// solve! with the Cross-Entropy loop over control sequences resident on the device (ce_device.hip: pets_sample_kernel / pets_update_kernel):
//   [control sequences from (mu_t, Sigma_t) -> S x K stochastic rollouts + sample means -> elites, smoothed update] x iter_max
// enqueued as ONE chain on the handle's stream; mu, Sigma and an error word come back through pinned memory behind ONE host wait.
// Rollout noise comes from the device generator (seed + iteration, as the host loop keys it); the control normals are the injected
// stream (read in place from pinned memory) or, zc == nullptr, drawn on the device.  Arithmetic = the host loop's: mu / Sigma bit for bit.
static bool pets_device_usable(rat_handle h, const rat_pets_solver *s, const double *zn) {
    return h->pets_device && !zn && s->num_control_samples >= 1 && s->num_control_samples <= PETS_DEV_MAX_S &&
           s->m <= 4 && s->N <= 1024 && s->num_elite >= 2 && s->num_elite <= s->num_control_samples && s->iter_max >= 1;
}
static rat_rc pets_solve_device(rat_handle h, rat_pets_solver *s, const double *x0, int32_t use_true_model, const double *zc, uint64_t seed) {
    HIPCHK(hipSetDevice(h->device));
    const int n = h->gn, m = h->gm, N = h->gN;
    const int64_t S = s->num_control_samples, K = s->num_trajectory_samples, IT = s->iter_max;
    const size_t ntraj = (size_t)S * K, nzc = (size_t)S * N * m, nmu = (size_t)N * m, nsg = (size_t)N * m * m;
    rat_rc rc;
    HIPCHK(hipStreamSynchronize(h->stream));                  // (a previous enqueue may still be reading the buffers)
    if ((rc = grow(&h->d_pin, &h->cap_pin, XSTR + (size_t)S * N * USTR))) return rc;
    if ((rc = grow(&h->d_ptraj, &h->cap_traj, ntraj))) return rc;
    if ((rc = grow(&h->d_pcost, &h->cap_cost, (size_t)S))) return rc;
    if ((rc = grow(&h->d_pmu, &h->cap_pmu, nmu))) return rc;
    if ((rc = grow(&h->d_psig, &h->cap_psig, nsg))) return rc;
    if (!h->d_perr) { HIPCHK(hipMalloc((void **)&h->d_perr, sizeof(double))); HIPCHK(hipMemset(h->d_perr, 0, sizeof(double))); }
    // pinned: [x0 (XSTR) | mu | Sigma | error word (as a double slot) | control normals of all iterations]
    const size_t off_mu = XSTR, off_sg = off_mu + nmu, off_err = off_sg + nsg, off_zc = off_err + 2, need = off_zc + (zc ? nzc * (size_t)IT : 0);
    if (need > h->cap_pzc) {
        if (h->h_pzc) (void)hipHostFree(h->h_pzc);
        h->h_pzc = nullptr; h->cap_pzc = 0;
        HIPCHK(hipHostMalloc((void **)&h->h_pzc, need * sizeof(double), hipHostMallocDefault));
        h->cap_pzc = need;
    }
    double *hp = h->h_pzc;
    memset(hp, 0, XSTR * sizeof(double));
    for (int i = 0; i < n; ++i) hp[i] = x0[i];
    memcpy(hp + off_mu, s->mu, nmu * 8);
    memcpy(hp + off_sg, s->Sigma, nsg * 8);
    if (zc) memcpy(hp + off_zc, zc, nzc * (size_t)IT * 8);
    const double *hp_dev = nullptr;
    HIPCHK(hipHostGetDevicePointer((void **)&hp_dev, hp, 0));
    const double *zc_dev = nullptr;
    if (zc) {                                                                         // the injected control normals: pulled over the link by one parallel launch
        if ((rc = grow(&h->d_pzu, &h->cap_zu, nzc * (size_t)IT))) return rc;          // (the rollouts draw their noise on the device: d_pzu is free)
        launch_pets_stage(hp_dev + off_zc, h->d_pzu, (long)(nzc * (size_t)IT), h->stream);
        zc_dev = h->d_pzu;
    }
    {                                                                                 // x0 | mu | Sigma in one launch, the error word zeroed by it
        PetsStage3 sa;
        sa.src[0] = hp_dev; sa.dst[0] = h->d_pin; sa.n[0] = (long)XSTR;
        sa.src[1] = hp_dev + off_mu; sa.dst[1] = h->d_pmu; sa.n[1] = (long)nmu;
        sa.src[2] = hp_dev + off_sg; sa.dst[2] = h->d_psig; sa.n[2] = (long)nsg;
        sa.zero_word = h->d_perr;
        launch_pets_stage3(sa, h->stream);
    }
    PetsArgs a;
    a.g = h->gen; a.x0 = h->d_pin; a.controls = h->d_pin + XSTR; a.S = S; a.K = K; a.use_true = use_true_model ? 1 : 0;
    a.zn = nullptr; a.zu = nullptr; a.traj0 = 0; a.traj_cost = h->d_ptraj; a.cost = h->d_pcost; a.wave16 = h->pets_wave16;
    // sample_1 | rollouts_1 | update_1 + sample_2 | rollouts_2 | ... | update_n : one bookkeeping launch between two rollout launches
    double *const ctrl = h->d_pin + XSTR;
    prof_begin(h, RAT_K_CE, S);
    launch_pets_step(h->d_pmu, h->d_psig, ctrl, h->d_pcost, (long)S, (int)s->num_elite, N, m, s->smoothing_factor, zc_dev, seed, 0, 0, 1,
                     h->d_perr, h->stream);
    prof_end(h);
    for (int64_t it = 0; it < IT; ++it) {
        a.seed = seed + (uint64_t)it;
        prof_begin(h, RAT_K_PETS, (int64_t)ntraj);
        launch_pets(a, h->stream);
        prof_end(h);
        const int more = (it + 1 < IT) ? 1 : 0;
        prof_begin(h, RAT_K_CE, S);
        launch_pets_step(h->d_pmu, h->d_psig, ctrl, h->d_pcost, (long)S, (int)s->num_elite, N, m, s->smoothing_factor,
                         (zc_dev && more) ? zc_dev + (size_t)(it + 1) * nzc : nullptr, seed, (int)(it + 1), 1, more, h->d_perr, h->stream);
        prof_end(h);
    }
    HIPCHK(hipGetLastError());
    // results back into the pinned area by the same pull kernel, the other way (device -> pinned host)
    double *hp_w = nullptr;
    HIPCHK(hipHostGetDevicePointer((void **)&hp_w, hp, 0));
    {                                                                                 // mu | Sigma | the error word (an 8-byte slot) by one launch
        PetsStage3 sa;
        sa.src[0] = h->d_pmu; sa.dst[0] = hp_w + off_mu; sa.n[0] = (long)nmu;
        sa.src[1] = h->d_psig; sa.dst[1] = hp_w + off_sg; sa.n[1] = (long)nsg;
        sa.src[2] = reinterpret_cast<const double *>(h->d_perr); sa.dst[2] = hp_w + off_err; sa.n[2] = 1;
        sa.zero_word = nullptr;
        launch_pets_stage3(sa, h->stream);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    int err = 0;
    memcpy(&err, hp + off_err, sizeof(int));
    if (err) return fail(RAT_ERR_ARG, "Sigma_t is not positive definite (MvNormal would throw)");
    memcpy(s->mu, hp + off_mu, nmu * 8);
    memcpy(s->Sigma, hp + off_sg, nsg * 8);
    s->iter_current = IT;
    return RAT_OK;
}

extern "C" rat_rc rat_pets_solve(rat_handle h, rat_pets_solver *s, const double *x0, int32_t use_true_model, const double *zc,
                                 const double *zn, const double *zu, uint64_t seed) {   // pets.jl:270-281
    if (!h || !s || !x0) return fail(RAT_ERR_ARG, "null");
    if (!h->have_gen) return fail(RAT_ERR_NO_PROBLEM, "rat_pets_problem_set was not called");
    if (s->N != h->gN || s->m != h->gm) return fail(RAT_ERR_ARG, "solver N / m do not match the problem");
    rat_pets_initialize(s);
    if (pets_device_usable(h, s, zn)) return pets_solve_device(h, s, x0, use_true_model, zc, seed);
    if (!zc) return fail(RAT_ERR_UNSUPPORTED, "rat_pets_solve: control normals are drawn on the device only by the device-resident loop (switch pets_device, "
                                              "no injected rollout noise, <= 1024 control samples)");
    const size_t nzc = (size_t)s->num_control_samples * s->N * s->m;
    const size_t ntraj = (size_t)s->num_control_samples * s->num_trajectory_samples;
    const size_t nzn = ntraj * s->N * h->gn, nzu = ntraj * s->N;
    while (s->iter_current < s->iter_max) {
        const int64_t it = s->iter_current;
        rat_rc rc = rat_pets_step(h, s, x0, use_true_model, zc + it * nzc, zn ? zn + it * nzn : nullptr, (zn && zu) ? zu + it * nzu : nullptr,
                                  seed + (uint64_t)it, nullptr, nullptr);
        if (rc) return rc;
    }
    return RAT_OK;
}
