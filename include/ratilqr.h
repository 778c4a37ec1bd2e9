/*
 * ratilqr.h -- C ABI of libratilqr_hip.so: the MI355X (gfx950) implementation of RATiLQR.jl's
 * iLEQG solve and of the Cross-Entropy loop over theta that wraps it.
 *
 * The reference (pure Julia) has no FFI boundary; its operator API for this path is the set of
 * exported functions in /root/reference/src/RATiLQR.jl:20-53.  Each entry point below names the
 * reference function it replaces (file:line into /root/reference/src).  The Julia-side `ccall`
 * bindings a maintainer would add are in INTEGRATION.md and julia/RATiLQRAMD.jl.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is fp64 (Float64), counters int32/int64;
 *   - matrices are COLUMN-MAJOR (Julia native), time is the slowest index: a Vector{Matrix}
 *     `L_array` of N (m x n) gains is the flat buffer L[i + m*j + m*n*t];
 *   - the caller owns every host buffer; the library copies in/out and retains no pointer after
 *     return (exception: the standard-normal stream registered with rat_ce_set_stream);
 *   - no exceptions cross the ABI: functions return a rat_rc (API misuse / HIP errors), and every
 *     trajectory carries a per-sample status (RAT_ST_*) with value = +Inf where the reference
 *     would have thrown (cross_entropy_bilevel_optimization.jl:161-165);
 *   - call from one host thread per handle; a handle owns one HIP device and one stream; a rat_multi owns one handle per
 *     device and is driven from one host thread as well (no callbacks, no thread-local state of the caller: @threadcall-safe);
 *   - user closures f/c/h/W cannot cross the ABI: problems are instances of compiled-in model
 *     families (rat_problem_desc.model).
 */
#ifndef RATILQR_H
#define RATILQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAT_VERSION 600

/* ---- return codes (API level) ---------------------------------------------------------------- */
typedef int32_t rat_rc;
#define RAT_OK               0
#define RAT_ERR_ARG          1   /* bad argument / option out of the reference's @assert ranges */
#define RAT_ERR_UNSUPPORTED  2   /* problem size or model outside the compiled kernels (n, m <= 32; power-law family n = m <= 4) */
#define RAT_ERR_HIP          3   /* HIP runtime error (see rat_last_error) */
#define RAT_ERR_NO_PROBLEM   4   /* rat_problem_set was not called */
#define RAT_ERR_STREAM_DRY   5   /* injected N(0,1) stream exhausted */
#define RAT_ERR_DIVERGED     6   /* a loop the reference would spin in forever was cut (App. B.11/B.15) */

/* ---- per-trajectory status (what the reference's exceptions become) -------------------------- */
#define RAT_ST_RUNNING          (-1)
#define RAT_ST_OK                 0  /* converged: d > d_current && mu <= mu_min  (ileqg.jl:642)            */
#define RAT_ST_M_NOT_PD_INIT      1  /* @assert isposdef(M) in initialize!        (ileqg.jl:234,440) -> Inf */
#define RAT_ST_M_NOT_PD_GAIN      2  /* @assert isposdef(M) in solve_approximate_dp! (ileqg.jl:366)  -> Inf */
#define RAT_ST_ITER_MAX           3  /* iter_max reached (ileqg.jl:648); value is valid                     */
#define RAT_ST_DOMAIN             4  /* DomainError / non-finite in rollout or linearisation         -> Inf */
#define RAT_ST_MU_DIVERGED        5  /* mu-restart loop cut                                         -> Inf */
#define RAT_ST_SINGULAR           6  /* reserved (SingularException)                                 -> Inf */
#define RAT_ST_LS_DIVERGED        7  /* line search cut after 4000 DP-failed candidates (App. B.5)  -> Inf */
#define RAT_ST_INTERNAL           8  /* a hand-over between the two workgroups of a sample timed out (never
                                        expected; reported instead of a device hang)                 -> Inf */

/* ---- model families --------------------------------------------------------------------------- */
#define RAT_MODEL_LQ        1  /* f = A x + B u + kappa x.^3 ; c_k, h quadratic (tables below)                */
#define RAT_MODEL_POWERLAW  2  /* f = x.^a + u.^b (n == m) ; c = cx sum(x.^p) + cu sum(u.^pu) ; h = pl_h   */

/* Replaces FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N)
 * (optimal_control_problems.jl:67-73).  Field order is shared with oracle/ratilqr_oracle.h. */
typedef struct rat_problem_desc {
    int32_t model;
    int32_t n, m, N;
    int32_t cost_tv;          /* 1: Q,R,P,qv,rv,q0 hold N entries (k = 0..N-1), else one entry */
    int32_t W_tv;             /* 1: W holds N entries, else one entry                            */
    const double *A;          /* n*n                */
    const double *B;          /* n*m                */
    const double *Q;          /* n*n [*N]  c_xx     */
    const double *R;          /* m*m [*N]  c_uu     */
    const double *P;          /* m*n [*N]  c_ux     */
    const double *qv;         /* n   [*N]           */
    const double *rv;         /* m   [*N]           */
    const double *q0;         /* 1   [*N]           */
    const double *Qf;         /* n*n  h_xx          */
    const double *qvf;        /* n                  */
    double q0f;
    double kappa;
    double pl_a, pl_b, pl_p, pl_pu, pl_cx, pl_cu, pl_h;
    const double *W;          /* n*n [*N]  noise covariance W(k) */
} rat_problem_desc;

/* Replaces the keyword arguments of ILEQGSolver(problem; ...)  (ileqg.jl:191-201). */
typedef struct rat_ileqg_opts {
    double mu_min, delta_0, lambda, d;
    int64_t iter_max;
    double eps_init, eps_min;
    int32_t adaptive_eps_init;
} rat_ileqg_opts;

typedef struct rat_handle_s *rat_handle;

int32_t     rat_version(void);
const char *rat_last_error(void);
void        rat_default_ileqg_opts(rat_ileqg_opts *o);                      /* defaults of ileqg.jl:191-194 */

/* Create a solver context on HIP device `device`.
 *   max_batch : largest number of theta-samples one batch call will carry (device buffers are sized once)
 *   spec_eps  : E >= 1, the LARGEST number of line-search step sizes eps_k = eps*lambda^k the library may evaluate
 *               speculatively per (sample, iteration); results are identical for every E.
 *               Speculation only pays where SIMDs would otherwise idle; on this device the sequential rule (E = 1) is at
 *               least as fast at every batch size, so a handle runs E = 1 unless the switch
 *               spec_force = 1 (rat_debug_set / RATILQR_SPEC_FORCE) asks for the requested width (the E > 1 kernels).
 * Replaces the ILEQGSolver constructor (ileqg.jl:191-208); option ranges are validated as its @asserts. */
rat_rc rat_create(const rat_ileqg_opts *opts, int32_t max_batch, int32_t spec_eps, int32_t device, rat_handle *out);
void   rat_destroy(rat_handle h);
rat_rc rat_set_ileqg_opts(rat_handle h, const rat_ileqg_opts *opts);

/* Upload a problem (tables are copied).  Replaces passing `problem` to every call.
 * Sizes: n <= 12, m <= 4 run on the MFMA kernels.  LQ-family problems up to n <= 32, m <= 32 are accepted too (the reference takes its
 * dimensions from the arrays, ileqg.jl:229) and run every entry point in general-size kernels; the power-law family beyond n = m = 4
 * and any larger problem return RAT_ERR_UNSUPPORTED. */
rat_rc rat_problem_set(rat_handle h, const rat_problem_desc *desc);

/* ---- the hot path ----------------------------------------------------------------------------- */

/* Batched iLEQG: one complete solve!(ileqg, problem, x0, u0; theta_i) per sample, all samples at once.
 * Replaces the fan-out of compute_value_worker (cross_entropy_bilevel_optimization.jl:144-167,186-191):
 * value[i] = solve!(...)[4], or +Inf where the reference would throw.  Optional outputs (may be NULL):
 * status[i] (RAT_ST_*), iters[i] (iLEQG iterations), ls_evals[i] (line-search candidates consumed by the
 * sequential rule of line_search!, ileqg.jl:504-581). x0[n], u0[m*N], theta[B]: host buffers. */
rat_rc rat_ileqg_solve_batch(rat_handle h, const double *x0, const double *u0, const double *theta, int64_t B,
                             double *value, int32_t *status, int32_t *iters, int32_t *ls_evals);

/* Same with theta / value / status / iters / ls_evals resident in device (HBM) memory; x0/u0 are taken
 * from the last rat_set_initial() call.  Asynchronous w.r.t. the host except for the per-round counter
 * read-back; returns after the batch has finished on the handle's stream. */
rat_rc rat_set_initial(rat_handle h, const double *x0, const double *u0);
rat_rc rat_ileqg_solve_batch_dev(rat_handle h, const double *theta_dev, int64_t B, double *value_dev,
                                 int32_t *status_dev, int32_t *iters_dev, int32_t *ls_evals_dev);

/* Single solve with the full policy returned.  Replaces solve!(ileqg, problem, x_0, u_array; theta)
 * (ileqg.jl:635-659): x[n*(N+1)], l[m*N], L[m*n*N], value, eps_history as (eps, new-current) pairs
 * (eps_hist holds 2*hist_cap doubles; *hist_n receives the number of pairs produced). */
rat_rc rat_ileqg_solve(rat_handle h, const double *x0, const double *u0, double theta,
                       double *x, double *l, double *L, double *value, int32_t *status, int32_t *iters,
                       double *eps_hist, int64_t hist_cap, int64_t *hist_n);

/* ---- individual operators (unit parity with test/ileqg_test.jl) -------------------------------- */

/* simulate_dynamics(problem, x_0, u_array)                     ileqg.jl:18-38   -> x[n*(N+1)]; *domain_fail = 1 on DomainError */
rat_rc rat_rollout_open(rat_handle h, const double *x0, const double *u, double *x, int32_t *domain_fail);
/* simulate_dynamics(problem, x_array, l_array, L_array)        ileqg.jl:62-87   -> x_new, u_new */
rat_rc rat_rollout_feedback(rat_handle h, const double *xbar, const double *l, const double *L,
                            double *x_new, double *u_new, int32_t *domain_fail);
/* integrate_cost(problem, x_array, u_array)                    ileqg.jl:115-124 */
rat_rc rat_integrate_cost(rat_handle h, const double *x, const double *u, double *cost);
/* simulate_dynamics(problem, x_0, u_array, rng)  ileqg.jl:44-55  (L == NULL: open loop, only the first column of x_nom is read) and
 * simulate_dynamics(problem, x_array, l_array, L_array, rng)  ileqg.jl:94-109  (affine policy u_k = l_k + L_k (x_k - x_nom_k)):
 * K independent Monte-Carlo rollouts x_{k+1} = f(x_k, u_k) + w_k, w_k ~ N(0, W(k)), drawn as chol_lower(W(k)) z_k (what
 * rand(rng, MvNormal(0, W)) computes).  z: [n x N x K] injected standard-normal draws (column-major, rollout slowest) or NULL for the
 * device generator (Philox4x32-10 keyed by seed; the reference's MersenneTwister stream is not reproducible).
 * Outputs (any may be NULL): x_out [n x (N+1) x K], u_out [m x N x K], cost_out [K] = integrate_cost of each rollout
 * (ileqg.jl:115-124; NaN where a rollout hit a DomainError), *domain_fail = 1 if any rollout did. */
rat_rc rat_rollout_noisy(rat_handle h, const double *x_nom, const double *l, const double *L, int64_t K,
                         const double *z, uint64_t seed, double *x_out, double *u_out, double *cost_out, int32_t *domain_fail);
/* approximate_model(problem, u_array, x_array)                 ileqg.jl:258-322
 * -> q[N+1], qv[n*(N+1)], Q[n*n*(N+1)], r[m*N], R[m*m*N], P[m*n*N], A[n*n*N], B[n*m*N], W[n*n*N] */
rat_rc rat_approximate_model(rat_handle h, const double *u, const double *x,
                             double *q, double *qv, double *Q, double *r, double *R, double *P,
                             double *A, double *B, double *W, int32_t *domain_fail);
/* solve_approximate_dp!(ileqg, approx; theta)                  ileqg.jl:341-406
 * in : the ApproximationResult arrays above (W is taken from the problem),  theta, mu, delta (in/out)
 * out: L[m*n*N], dl[m*N], updated mu / delta (regularisation restarts), status (0 / RAT_ST_M_NOT_PD_GAIN / ...),
 *      and the DynamicProgrammingResult dumps (any may be NULL): s[N+1], sv[n*(N+1)], S[n*n*(N+1)],
 *      g[m*N], G[m*n*N], H[m*m*N]. */
rat_rc rat_dp_gain_sweep(rat_handle h, const double *q, const double *qv, const double *Q, const double *r,
                         const double *R, const double *P, const double *A, const double *B,
                         double theta, double *mu, double *delta, double *L, double *dl, int32_t *status,
                         double *s, double *sv, double *S, double *g, double *G, double *H);
/* solve_approximate_dp(approx, L_array, dl_array; theta, mu)    ileqg.jl:412-465 ; dl may be NULL (zeros);
 * *status = 0 or RAT_ST_M_NOT_PD_GAIN (the @assert at :440). */
rat_rc rat_dp_policy_eval(rat_handle h, const double *q, const double *qv, const double *Q, const double *r,
                          const double *R, const double *P, const double *A, const double *B,
                          const double *L, const double *dl, double theta, double mu, int32_t *status,
                          double *s, double *sv, double *S, double *g, double *G, double *H);

/* Batched forms of the two sweeps on CALLER-SUPPLIED tiles -- the batch path of problems whose f, c, h are arbitrary host closures
 * (optimal_control_problems.jl:67-73; ileqg.jl:265-273, :302-311): the host rolls out and linearises every sample, the device runs the
 * B Riccati sweeps of one CE batch in one launch (one wavefront per sample, the solver's own kernels).  Every array holds B consecutive
 * ApproximationResults / gain histories in the single-sample layouts above (sample slowest); theta[B], mu[B], delta[B] per sample.
 *   rat_dp_gain_sweep_batch  : solve_approximate_dp! per sample (mu restarts inside): in/out mu, delta; out L, dl, status[b]
 *   rat_dp_policy_eval_batch : solve_approximate_dp with dl = nothing per sample (initialize!, line-search candidates): out value[b] =
 *                              s_array[1] (+Inf and status RAT_ST_M_NOT_PD_GAIN where the reference's @assert fires) */
rat_rc rat_dp_gain_sweep_batch(rat_handle h, int64_t B, const double *q, const double *qv, const double *Q, const double *r,
                               const double *R, const double *P, const double *A, const double *Bm, const double *theta,
                               double *mu, double *delta, double *L, double *dl, int32_t *status);
rat_rc rat_dp_policy_eval_batch(rat_handle h, int64_t B, const double *q, const double *qv, const double *Q, const double *r,
                                const double *R, const double *P, const double *A, const double *Bm, const double *L,
                                const double *theta, const double *mu, double *value, int32_t *status);

/* ---- Cross-Entropy loop over theta (RAT iLQR) -------------------------------------------------- */

/* Replaces CrossEntropyBilevelOptimizationSolver (cross_entropy_bilevel_optimization.jl:70-127).
 * The struct is caller-owned plain data: mu_init/sigma_init persist across solves as in the reference. */
typedef struct rat_ce_solver {
    /* parameters */
    int64_t num_samples, num_elite, iter_max;
    double  lambda;
    int32_t use_theta_max;
    /* mutable state */
    double  mu_init, sigma_init, mu, sigma, theta_max, theta_min;
    int64_t iter_current;
    /* bookkeeping (not in the reference) */
    int64_t n_solves, n_redraws;
    int64_t n_final_retries;     /* times the final solve of solve! failed and theta_opt was lowered by sigma (:410-413) */
} rat_ce_solver;

void   rat_ce_default(rat_ce_solver *c);                                     /* ctor defaults :100-127 */
void   rat_ce_initialize(rat_ce_solver *c);                                  /* initialize!   :133-138 */

/* Source of randomness replacing `rng::AbstractRNG`: theta = mu + sigma*z with z drawn in order from a
 * standard-normal stream.  Either inject one (parity runs; the pointer must stay valid while in use) or
 * seed the built-in generator (SplitMix64-seeded xoshiro256++ with Box-Muller; documented in DESIGN.md). */
rat_rc rat_ce_set_stream(rat_handle h, const double *z, int64_t nz);
rat_rc rat_ce_seed(rat_handle h, uint64_t seed);
int64_t rat_ce_stream_pos(rat_handle h);

/* get_positive_samples(mu, sigma, num_samples, rng)             :233-246 */
rat_rc rat_ce_get_positive_samples(rat_handle h, double mu, double sigma, int64_t num, double *theta);
/* compute_cost(ce_solver, problem, x, u_array, theta_array, kl_bound)  :173-195  (cost = value + kl/theta) */
rat_rc rat_ce_compute_cost(rat_handle h, const double *x0, const double *u0, const double *theta, int64_t B,
                           double kl_bound, double *cost);
/* compute_cost with theta_dev / cost_dev resident in device (HBM) memory; x0/u0 from the last rat_set_initial() call.
 * cost = value + kl_bound / theta (:193), +Inf for samples whose solve failed (:163-165).  Returns after the batch has finished. */
rat_rc rat_ce_compute_cost_dev(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev);
/* the same, stream-ordered: returns once the batch is enqueued on rat_stream(h) (single-launch path; otherwise it behaves like
 * rat_ce_compute_cost_dev).  theta_dev / cost_dev must stay valid, and cost_dev unread, until work ordered after it on that stream
 * (or a synchronisation of it) has passed -- lets a caller chain batch -> cost all-gather -> next batch without host round trips. */
rat_rc rat_ce_compute_cost_enqueue(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev);
/* the same with the per-sample status (RAT_ST_*), iteration count and line-search evaluation count written beside the costs (device
 * pointers, any of the three may be NULL): what a rank contributes to the cost + status all-gather of a sharded CE batch */
rat_rc rat_ce_compute_cost_enqueue_ex(rat_handle h, const double *theta_dev, int64_t B, double kl_bound, double *cost_dev,
                                      int32_t *status_dev, int32_t *iters_dev, int32_t *ls_evals_dev);
/* The bookkeeping half of step! (:291-334) for hosts that evaluate costs themselves (multi-GPU: the
 * host all-gathers cost shards between rat_ce_draw and rat_ce_update).
 *   rat_ce_draw   : theta[num_samples] for the current iteration (uses mu_init/sigma_init in iteration 1)
 *   rat_ce_update : consumes the costs; *redraw = 1 when the reference would loop and redraw (:293-298, :306) */
rat_rc rat_ce_begin_step(rat_ce_solver *c);
rat_rc rat_ce_draw(rat_handle h, const rat_ce_solver *c, double *theta);
rat_rc rat_ce_update(rat_ce_solver *c, const double *theta, const double *cost, int32_t *redraw);
/* rat_ce_update carried out by the update kernel of the device-resident loop (what rat_ce_solve runs between two batches) on
 * host-supplied thetas / costs: same arithmetic, same elite order -- sort(by = cost) under Julia's isless (NaN last, -0.0 before
 * +0.0, ties in input order; :326-328).  num_samples <= 1024. */
rat_rc rat_ce_update_dev(rat_handle h, rat_ce_solver *c, const double *theta, const double *cost, int32_t *redraw);
/* handle-free form of rat_ce_draw over an explicit stream (pure host code; usable before any device exists):
 * consumes z[*zpos..] and advances *zpos. */
rat_rc rat_ce_draw_stream(const rat_ce_solver *c, const double *z, int64_t nz, int64_t *zpos, double *theta);
/* step!(ce_solver, problem, x, u_array, kl_bound, rng)          :252-335 ; theta_out/cost_out may be NULL */
rat_rc rat_ce_step(rat_handle h, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                   double *theta_out, double *cost_out);
/* solve!(ce_solver, problem, x_0, u_array, rng; kl_bound)       :364-415 */
rat_rc rat_ce_solve(rat_handle h, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                    double *theta_opt, double *x, double *l, double *L, double *value,
                    double *theta_min, double *theta_max);

/* ---- RAT iLQR++: Nelder-Mead over theta ---------------------------------- */

/* Replaces NelderMeadBilevelOptimizationSolver (nelder_mead_bilevel_optimization.jl:72-128).  Caller-owned plain data.
 * c_high / c_low are Union{Nothing,Float64} in the reference and are NOT reset by initialize! (:164-168): they persist
 * across solve! calls, as do theta_high_init / theta_low_init when they were shrunk (:290-303).  Reproduced as is. */
typedef struct rat_nm_solver {
    double  alpha, beta, gamma, eps, lambda;
    int64_t iter_max;
    double  theta_high_init, theta_low_init;
    int64_t iter_current;
    double  theta_high, theta_low;
    int32_t has_c_high, has_c_low;
    double  c_high, c_low;
    int64_t n_solves;           /* iLEQG solves the SEQUENTIAL algorithm would have made (bookkeeping) */
    int64_t n_batches;          /* batched device calls actually made */
} rat_nm_solver;

void   rat_nm_default(rat_nm_solver *s);                                     /* ctor defaults :102-128 */
void   rat_nm_initialize(rat_nm_solver *s);                                  /* initialize!   :164-168 */
/* compute_cost_worker(nm_solver, problem, x, u_array, theta, kl_bound)        :134-158 */
rat_rc rat_nm_compute_cost(rat_handle h, const double *x0, const double *u0, double theta, double kl_bound, double *cost);
/* step! :174-252.  Every theta the sequential logic can ask for -- this iteration's reflection, expansion, two possible
 * contraction and two possible shrink points and, as far as the handle's max_batch allows (80 samples), the six points of
 * each state the iteration can end in -- is solved ahead in ONE batch (a batch of <= 512 samples takes one solve's time); the
 * reflect / expand / contract / shrink decisions are then replayed on the host against the table of (theta, cost), so the
 * outcome is the sequential one and the next call usually needs no device call.  The table is kept between calls while
 * (problem, x0, u0, kl_bound) are unchanged.  Switch nm_depth (rat_debug_set) limits the speculation. */
rat_rc rat_nm_step(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound);
/* solve!(nm_solver, problem, x_0, u_array; kl_bound)                           :276-352 ; *status = final iLEQG status
 * (a failure there is an uncaught exception in the reference).  Both initial vertices and the first two iterations under
 * either ordering go into the first device call (158 samples; three iterations, 1022 samples, on a handle that large), later calls
 * cover two iterations each, and the final
 * solve at theta_opt is read out of the last batch's device state instead of being run again. */
rat_rc rat_nm_solve(rat_handle h, rat_nm_solver *s, const double *x0, const double *u0, double kl_bound,
                    double *theta_opt, double *x, double *l, double *L, double *value, int32_t *status);

/* ---- PETS: cross-entropy over control sequences with stochastic rollouts ---------- */

/* Replaces FiniteHorizonGenerativeOptimalControlProblem(f_stochastic, c, h, N) (optimal_control_problems.jl:126-131) by a
 * device model family:  f_stochastic(x, u, rng, use_true_model) = A x + B u + kappa x.^3 + w
 *   w : noise_kind 0 -> N(nmean, nchol nchol') ; 1 -> uniform on [nlo, nhi)^n (test/pets_test.jl:15)
 *   use_true_model : with probability tw2 the noise is N(tmean2, tchol2 tchol2') instead (2-component mixture of the docs
 *                    example, optimal_control_problems.jl:103-110); tw2 = 0 disables it
 *   c(k, x, u) = the LQ quadratic form of `lq` (time-varying tables allowed) + l1u * sum(abs.(u)) ; h quadratic (lq.Qf ...). */
typedef struct rat_gen_problem_desc {
    rat_problem_desc lq;        /* model must be RAT_MODEL_LQ; lq.W is ignored (may be NULL) */
    double l1u;
    int32_t noise_kind;
    const double *nmean;        /* n        */
    const double *nchol;        /* n*n column-major, lower triangular */
    double nlo, nhi;
    double tw2;
    const double *tmean2;       /* n        */
    const double *tchol2;       /* n*n      */
} rat_gen_problem_desc;

/* Replaces CrossEntropyDirectOptimizationSolver (pets.jl:36-68).  Caller-owned; the mu and Sigma pointers are caller-owned buffers
 * [N][m] and [N][m*m] (column-major blocks, time slowest). */
typedef struct rat_pets_solver {
    int64_t num_control_samples, num_trajectory_samples, num_elite, iter_max;
    double  smoothing_factor;
    int64_t N, m, iter_current;
    double *mu_init, *Sigma_init, *mu, *Sigma;
} rat_pets_solver;

rat_rc rat_pets_problem_set(rat_handle h, const rat_gen_problem_desc *desc);
void   rat_pets_initialize(rat_pets_solver *s);                               /* initialize!  pets.jl:70-74 */
/* compute_cost_serial(direct_solver, problem, x, control_sequence_array, rng, use_true_model)  pets.jl:128-157
 *   controls[S][N][m] (time-major, m fastest), cost[S] = mean over K stochastic rollouts of sum c + h.
 * Randomness, serial semantics: trajectory j = ii*K + kk consumes zn[(j*N + t)*n .. +n) at step t (N(0,1) draws for Gaussian
 * noise, U[0,1) draws for uniform noise) and zu[j*N + t] (mixture choice; may be NULL when tw2 = 0).  zn = NULL selects the
 * device generator (Philox4x32-10 keyed by `seed`, counter = (trajectory, step / 2, lane); both Box-Muller outputs are used, for steps 2i and 2i + 1): statistical parity only. */
rat_rc rat_pets_compute_cost(rat_handle h, const double *x0, const double *controls, int64_t S, int64_t K,
                             int32_t use_true_model, const double *zn, const double *zu, uint64_t seed, double *cost);
/* draw the control sequences of one step! (pets.jl:206-216): controls[ii][t] = mu_t + chol(Sigma_t) * zc[(ii*N + t)*m ..] */
rat_rc rat_pets_sample_controls(const rat_pets_solver *s, const double *zc, double *controls);
/* get_elite_samples + compute_new_distribution (pets.jl:159-191); elite_idx[num_elite] may be NULL */
rat_rc rat_pets_update(rat_pets_solver *s, const double *controls, const double *cost, int64_t *elite_idx);
/* step! (pets.jl:193-245): sample, evaluate, elites, smoothed update.  zc: S*N*m normals; zn/zu/seed as above. */
rat_rc rat_pets_step(rat_handle h, rat_pets_solver *s, const double *x0, int32_t use_true_model, const double *zc,
                     const double *zn, const double *zu, uint64_t seed, double *controls_out, double *cost_out);
/* solve! (pets.jl:270-281): iter_max steps from (mu_init, Sigma_init); streams hold iter_max consecutive step blocks
 * (zn NULL -> device generator with seed + iteration).  With zn NULL and <= 1024 control samples the whole loop stays on the device
 * (switch pets_device): sampling, rollouts, elite selection and the smoothed update are one enqueue chain, one host wait per solve!;
 * mu / Sigma equal the host loop's bit for bit.  zc NULL (device-resident loop only): the control normals are drawn on the device too. */
rat_rc rat_pets_solve(rat_handle h, rat_pets_solver *s, const double *x0, int32_t use_true_model, const double *zc,
                      const double *zn, const double *zu, uint64_t seed);

/* ---- several devices behind one object -----------------------------------------------------------------
 * Replaces the process fan-out of compute_cost (cross_entropy_bilevel_optimization.jl:180-192: `@sync ... @async remotecall_fetch(
 * compute_value_worker, 2 + mod(i, nprocs - 1), ...)` over `addprocs` workers) and of the PETS cost (pets.jl:108-124): ONE host
 * thread drives n_devices GPUs.  theta-samples are split in contiguous blocks (rat_shard_bounds), every device solves its block in
 * one launch on its own HIP stream, and ONE ncclAllGather (RCCL over xGMI) of the per-sample costs, ordered on those streams, leaves
 * cost[B] on every device; device 0's copy returns to the host.  Elite selection stays host arithmetic (rat_ce_update), the final
 * solve at theta_opt runs on device 0.  Results do not depend on n_devices. */
typedef struct rat_multi_s *rat_multi;
/* contiguous block [lo, hi) of `rank` among `world` (blocks differ by at most one sample; device-free) */
rat_rc  rat_shard_bounds(int64_t B, int32_t world, int32_t rank, int64_t *lo, int64_t *hi);
/* devices: n_devices distinct HIP device indices, or NULL for 0 .. n_devices-1.  max_batch is the whole CE batch. */
rat_rc  rat_create_multi(const rat_ileqg_opts *opts, int32_t max_batch, int32_t spec_eps, int32_t n_devices,
                         const int32_t *devices, rat_multi *out);
void    rat_multi_destroy(rat_multi m);
int32_t rat_multi_n_devices(rat_multi m);
rat_handle rat_multi_handle(rat_multi m, int32_t i);        /* the single-device handle of device i (e.g. rat_ce_set_stream on i = 0) */
int32_t rat_multi_uses_rccl(rat_multi m);                   /* 1 when the costs travel through ncclAllGather */
int64_t rat_multi_allgathers(rat_multi m);                  /* collectives issued so far (one per compute_cost batch) */
rat_rc  rat_multi_problem_set(rat_multi m, const rat_problem_desc *desc);
rat_rc  rat_multi_set_initial(rat_multi m, const double *x0, const double *u0);
/* compute_cost (:173-195) on all devices; x0/u0 may be NULL (keep the last rat_multi_set_initial) */
rat_rc  rat_multi_ce_compute_cost(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B,
                                  double kl_bound, double *cost);
/* the same with the gathered per-sample status (RAT_ST_*), iLEQG iteration count and line-search evaluation count of every shard
 * (the all-gather carries cost + status); any of the three may be NULL */
rat_rc  rat_multi_ce_compute_cost_ex(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B,
                                     double kl_bound, double *cost, int32_t *status, int32_t *iters, int32_t *ls_evals);
/* rat_ileqg_solve_batch (compute_value_worker over a batch, :144-167) on all devices: value (+Inf for failures) and the per-sample counters */
rat_rc  rat_multi_ileqg_solve_batch(rat_multi m, const double *x0, const double *u0, const double *theta, int64_t B,
                                    double *value, int32_t *status, int32_t *iters, int32_t *ls_evals);
/* 1 when the devices are logical (test hook RATILQR_MULTI_LOGICAL=1: device index d runs on physical device d mod the visible count
 * and the all-gather is carried out by stream-ordered device copies: the G > 1 code on a one-GPU box) */
int32_t rat_multi_is_logical(rat_multi m);
/* step! (:252-335) / solve! (:364-415) with the cost evaluation on all devices; draws come from rat_multi_handle(m, 0) */
rat_rc  rat_multi_ce_step(rat_multi m, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                          double *theta_out, double *cost_out);
rat_rc  rat_multi_ce_solve(rat_multi m, rat_ce_solver *c, const double *x0, const double *u0, double kl_bound,
                           double *theta_opt, double *x, double *l, double *L, double *value,
                           double *theta_min, double *theta_max);
/* PETS on all devices (pets.jl:100-126, the `remotecall_fetch(compute_cost_worker, ...)` fan-out :108-124): the S control samples in
 * contiguous blocks, all K stochastic rollouts of a sample on one device, costs straight to the host; arguments as rat_pets_compute_cost.
 * Costs do not depend on n_devices (injected noise is addressed by global sample index, the device generator by global trajectory). */
rat_rc  rat_multi_pets_problem_set(rat_multi m, const rat_gen_problem_desc *desc);
rat_rc  rat_multi_pets_compute_cost(rat_multi m, const double *x0, const double *controls, int64_t S, int64_t K,
                                    int32_t use_true_model, const double *zn, const double *zu, uint64_t seed, double *cost);

/* ---- execution path of the batched solves of a handle ------------------------------------------
 * Results are identical on every path (tested bit for bit); AUTO picks by batch size, speculation width E and the device's CU count:
 *   E = 1: BLOCK up to 2 n_cu samples (a sample's evaluation and gain recursions side by side on two SIMDs), FUSED beyond (in-wave
 *          pairing; batches beyond one sample per SIMD run that kernel in generations of workgroups);
 *   E = 2 / 4 / 8: BLOCK while the batch fits one generation of workgroups (n_cu * floor(8 / (E + 1)) samples; E = 8: n_cu), ROUNDS beyond;
 *   any other E, and the operator entry points: ROUNDS.
 * rat_set_path fixes the path of the handle (RAT_ERR_UNSUPPORTED when the handle's E has no such kernel) and overrides the `block` /
 * `fused` switches below; RAT_PATH_AUTO returns to what those switches say (a handle created on the round-based path stays there). */
#define RAT_PATH_AUTO   0
#define RAT_PATH_ROUNDS 1   /* one launch per phase, rounds polled by the host */
#define RAT_PATH_FUSED  2   /* one persistent wavefront per sample: whole solve! in one launch (E = 1) */
#define RAT_PATH_BLOCK  3   /* one workgroup per sample: whole solve! in one launch (E = 1, 2, 4, 8) */
rat_rc  rat_set_path(rat_handle h, int32_t path);
/* the path (RAT_PATH_ROUNDS / _FUSED / _BLOCK; 4 = general-size kernel) a batch of B samples takes on this handle, or -1 */
int32_t rat_get_path(rat_handle h, int64_t B);

/* ---- execution switches: ONE entry point for tests, A/B tools and bench.py's contract leg ------------------------------------------
 * Results never depend on a switch, except `wdiag` and `block_acl` (another rounding order: ~1e-13 relative).  At rat_create every switch also takes
 * the value of the environment variable RATILQR_<KEY IN CAPITALS> when that is set (the library reads no other environment variable
 * besides RATILQR_MULTI_LOGICAL / RATILQR_MULTI_FORCE_RCCL of rat_create_multi).  A switch that changes the HBM layout of the handle's
 * state (`fused`, `dual`, `speculate` on an E = 1 handle) re-lays it: give rat_set_initial again, as after rat_set_path.
 *   key             values   meaning (default)
 *   fused           0 / 1    E = 1: single-launch solves (1) or one launch per phase, "round-based path" (0)            (1)
 *   block           -1/0/1   workgroup-per-sample kernel: by batch size (-1), never (0), whenever it exists (1)          (-1)
 *   block_max_b     B        E = 1: largest batch the workgroup-per-sample kernel takes under block = -1                 (2 n_cu)
 *   block_shape     0 / 1    two-wave workgroups padded to one wave per SIMD with ticketed SIMD pairs                   (1)
 *   block_helpers   0 / 1    spare waves of a padded workgroup linearise (one workgroup per CU)                          (1)
 *   block_acl       0 / 1    E = 1 workgroup-per-sample kernel: closed-loop rollouts in deviation form (3 MFMAs on the recursion's
 *                            chain; values agree with the other paths to rounding, ~1e-15, not bit for bit; opt-in)      (0)
 *   fused_dual      (read)   policy evaluation + following gain sweep as two recursions of one wavefront: always on (the separate-sweep
 *                            instantiations of rounds 1-5 were retired in round 6; writes are ignored)                    (1)
 *   fused_occ2      B0       batches of >= B0 samples: the 256-register one-recursion kernel, two samples per SIMD       (0 = never;
 *                            -1, the default: LQ-family batches of more samples than the device has SIMDs)
 *   spec_force      0 / 1    run the speculation width rat_create was given (kernels for E = 2, 4, 8 in one launch, any E on the
 *                            round-based path) instead of the sequential rule; re-lays the handle's state like `fused`         (0)
 *   spec_width      (read)   the width the handle runs: 1, or rat_create's spec_eps under spec_force
 *   prune           0 / 1    round-based path, E > 1, tile-free candidates: the evaluations of candidates 1 .. E-1 of a sample stop once
 *                            candidate 0 is known to be the line search's choice (identical outputs)                      (1)
 *   wide16          0 / 1    general sizes with 12 <= n <= 16, m <= 4 (beyond the 12 + 4 tile): sweeps and rollouts of the solve kernel in
 *                            registers on the matrix pipe (wide16.h); 0 = the general LDS sweep.  rat_debug_get: 1 only where the
 *                            problem set on the handle really runs that form                                              (1)
 *   wide32          0 / 1    every other general size (n <= 32, m <= 32): the same in block form on 16 x 16 tiles, tables as register images
 *                            (wide32.h: one wavefront per SIMD instead of one per compute unit); 0 = the general LDS sweep      (1)
 *   init_share      0 / 1    initialize!'s rollout (independent of theta) rolled out once per (x_0, u_array) and copied   (1)
 *   init_lazy       0 / 1    ... except for the FIRST batch on a new (x_0, u_array) while a sample has a compute unit to itself (<= n_cu samples):
 *                            the samples roll it out inside the solve kernel and the shared rollout's launch (19 us) stays off the
 *                            critical path of one-shot callers -- receding-horizon solves, a single rat_ileqg_solve               (1)
 *   materialize     0 / 1    one-wavefront-per-sample kernel, LQ family, time-invariant cost: tile records written by the
 *                            rollouts and loaded by the sweeps (SURVEY 8d's wording) instead of formed in registers      (0)
 *   fly             0 / 1    round-based path, E > 1: line-search candidates without tile records                         (1)
 *   fly_multi       0 / 1    ... and all candidates of a sample rolled out by one wavefront                               (1)
 *   dual            0 / 1    round-based path: candidate 0 paired with the next gain sweep in one wavefront              (E > 1)
 *   speculate       0 / 1    round-based path: speculative gain sweeps on a second stream                                (0)
 *   nm_depth        0 .. 3   Nelder-Mead speculation: 0 the six vertices of the iteration per device call; 1 also the two current vertices (the
 *                            final solve is read out of the last batch) and both initial vertices with the first iteration in one call; 2 also
 *                            the vertices of the iteration after (two iterations per device call); 3 also a third iteration in the first call
 *                            of rat_nm_solve (handles of >= 1022 samples).  Results do not depend on it  (3)
 *   pets_wave16     0 .. 3   PETS rollouts: 0 four per wavefront; 1 sixteen per wavefront as MFMA columns, the noise drawn by three generator
 *                            wavefronts per workgroup while the launch is small (<= 1536 wavefronts), by the recursion's own beyond; 2 never
 *                            split; 3 always split.  1-3 are bit-identical, 0 agrees to rounding  (1)
 *   ce_device       0 / 1    rat_ce_solve keeps the CE loop on the device: draw / update kernels, one host wait per solve!        (1)
 *   pets_device     0 / 1    rat_pets_solve keeps the CE loop over control sequences on the device (one host wait per solve!)      (1)
 *   block_psw       0 / 1    E = 1 batches of at most one sample per compute unit (LQ family): the workgroup-per-sample solve with every Riccati
 *                            sweep TIME-PARALLEL over the sample's four SIMDs (solve_block_psw_kernel, csrc/psweep.h); status / iteration /
 *                            line-search counts as on every other path, values equal to rounding (~1e-15), not bit for bit          (1)
 *   psw_acl         0 / 1    ... its closed-loop rollouts in deviation form (3 MFMAs on the recursion's chain)                      (1)
 *   psw_duo         0 / 1    ... with TWO workgroups (compute units) per sample while the batch leaves half the device dark (<= n_cu / 2
 *                            samples): the policy evaluations as four-wave teams on one, every gain sweep as a four-wave team on the
 *                            other, hand-overs through the XCD's L2 (counts identical, values to rounding: other segment cuts)         (1)
 *   psw_duo_count   (read)   samples of this handle that have run that way so far (rat_debug_set clears it)
 *   psw_prl         0 / 1    ... and the candidate's closed-loop rollout (simulate_dynamics, ileqg.jl:62-87) TIME-PARALLEL over the four
 *                            wavefronts where the deviation from the nominal trajectory is affine (kappa == 0, time-invariant cost,
 *                            N >= 16): four segments, the deviation at each cut from the composed maps of the segments before it
 *                            (rollprl_body; counts identical, values to ~1e-15 against the one-wave recursion)                         (1)
 *   prl_elem, prl_hop, prl_epi   its cost model in hundredths of an ordinary rollout step -- one step of a segment map, one hop, the
 *                            terminal tile: where the three cuts go (any model gives the same results)                      (45, 90, 100)
 *   prl_cuts        (read)   the cuts of the last launch that ran it: cut_1 | cut_2 << 16 | cut_3 << 32 (0: it did not apply)
 *   psweep          0, 2..4  the batched sweep operators (rat_dp_gain_sweep_batch / rat_dp_policy_eval_batch) run the TIME-PARALLEL sweep:
 *                            that many wavefronts per trajectory over that many + 1 horizon segments (csrc/psweep.h); results agree with the
 *                            sequential sweep to rounding (not bit for bit); values above 4 mean 4 (one wave per SIMD)             (0)
 *   psw_hop, psw_hop_e, psw_comp   its cost model in hundredths of an ordinary step -- one hop of a gain sweep, one hop of an evaluation, one
 *                            element step: where the segment cuts go                                                      (120, 140, 125)
 *   wdiag           0 / 1    diagonal time-invariant W: inv(W) folded into M^-1's operand (takes effect at the next rat_problem_set) (1) */
rat_rc  rat_debug_set(rat_handle h, const char *key, int64_t value);
rat_rc  rat_debug_get(rat_handle h, const char *key, int64_t *value);      /* the EFFECTIVE value on this handle */

/* ---- measurement hooks (bench.py) -------------------------------------------------------------- */
#define RAT_K_ROLLOUT   0
#define RAT_K_LINEARIZE 1
#define RAT_K_SWEEP_EVAL 2
#define RAT_K_SWEEP_GAIN 3
#define RAT_K_SELECT    4
#define RAT_K_SWEEP_INIT 5   /* open-loop policy evaluation of initialize! (no gains read) */
#define RAT_K_SWEEP_DUAL 6   /* fused wavefront: policy evaluation + the next step!'s gain sweep over one pass of the tiles */
#define RAT_K_SOLVE_FUSED 7  /* one persistent wavefront per sample runs the whole solve! (E = 1): every phase above in one launch */
#define RAT_K_SOLVE_BLOCK 8  /* one workgroup per sample runs the whole solve!: a wavefront per line-search candidate + a gain-sweep wavefront */
#define RAT_K_SOLVE_WIDE  9  /* general-size solve kernel (n <= 32, m <= 32 beyond the 12 + 4 tile): a workgroup per sample, whole solve! */
#define RAT_K_PETS       10  /* PETS stochastic rollouts (pets_rollout_kernel + the per-sample mean) */
#define RAT_K_CE         11  /* draw + update kernels of the device-resident Cross-Entropy loops: rat_ce_solve (one workgroup each), rat_pets_solve (one per time step) */
#define RAT_K_COUNT     12
/* When enabled, kernel launches are bracketed by HIP events on the handle's stream.
 * on = 0: off; on = 1: every kernel kind; otherwise on = (mask << 1) | 1 with bit k of mask selecting kind RAT_K_k.
 * Launches of a surplus round (no live sample left) are not counted. */
rat_rc rat_profile_enable(rat_handle h, int32_t on);
rat_rc rat_profile_reset(rat_handle h);
/* launches[k], trajectories[k] (units processed), total_ms[k] for k < RAT_K_COUNT */
rat_rc rat_profile_get(rat_handle h, int64_t *launches, int64_t *trajectories, double *total_ms);
/* the handle's HIP stream as an opaque pointer (hipStream_t), and its device index */
void  *rat_stream(rat_handle h);
/* bytes of one trajectory's tile bundle as laid out in HBM, and of the per-trajectory L/x/u arrays */
rat_rc rat_layout_info(rat_handle h, int64_t *tile_bytes, int64_t *L_bytes, int64_t *x_bytes, int64_t *u_bytes);

#ifdef __cplusplus
}
#endif
#endif
