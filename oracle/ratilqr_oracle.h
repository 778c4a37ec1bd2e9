/*
 * ratilqr_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C fp64 restatement of the reference's iLEQG solver (src/ileqg.jl) and of the
 * Cross-Entropy bilevel loop over theta (src/cross_entropy_bilevel_optimization.jl) of
 * StanfordMSL/RATiLQR.jl.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product library (libratilqr_hip.so) never does.
 *
 * Pinning: the reference is pure Julia and no `julia` exists in the build container, so
 * the oracle cannot be diffed against reference output.  It is pinned by the reference's
 * own known-answer tests (test/ileqg_test.jl, test/cross_entropy_bilevel_optimization_test.jl;
 * K1..K15 of SURVEY.md section 8c), re-stated in tests/test_oracle_*.py.
 * RNG streams (Julia MersenneTwister + ziggurat randn) are NOT reproduced: CE parity is
 * defined on injected standard-normal streams ("parity unpinned" for RNG only).
 *
 * Third-party arithmetic the reference leans on and how it is restated here:
 *   ForwardDiff 0.10.12 (Manifest.toml:81)  -> analytic derivatives of the compiled-in model families
 *   LinearAlgebra/LAPACK (Julia 1.5 stdlib) -> inv (LU), isposdef (Cholesky), Symmetric right-division
 *                                              (Cholesky solve here; Bunch-Kaufman there), `\` (LU),
 *                                              logdet (LU)
 *   Distributions 0.24.2 Normal sampling    -> theta = mu + sigma*z with z from an injected N(0,1) stream
 *
 * Matrix layout everywhere: column-major (Julia native), time is the slowest index.
 */
#ifndef RATILQR_ORACLE_H
#define RATILQR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODEL_LQ        1  /* f = A x + B u + kappa*x.^3 ; quadratic c_k, h */
#define ORC_MODEL_POWERLAW  2  /* f = x.^a + u.^b ; c = cx*sum(x.^p) + cu*sum(u.^pu) ; h = const (test/ileqg_test.jl:151-155) */

/* Status codes shared with the GPU library (include/ratilqr.h). */
#define ORC_OK                 0
#define ORC_ERR_M_NOT_PD_INIT  1  /* @assert isposdef(M) failed inside initialize!   (ileqg.jl:234,440) */
#define ORC_ERR_M_NOT_PD_GAIN  2  /* @assert isposdef(M) failed in solve_approximate_dp! (ileqg.jl:366) */
#define ORC_ITER_MAX           3  /* finished by iter_max (ileqg.jl:648) -- value is valid */
#define ORC_ERR_DOMAIN         4  /* DomainError / non-finite in rollout or linearisation */
#define ORC_ERR_MU_DIVERGED    5  /* mu restart loop did not terminate (reference would spin) */
#define ORC_ERR_SINGULAR       6  /* SingularException from an LU solve */

typedef struct orc_problem {
    int32_t model;
    int32_t n, m, N;
    int32_t cost_tv;          /* 1: Q,R,P,qv,rv,q0 hold N entries (k = 0..N-1); 0: one entry reused */
    int32_t W_tv;             /* 1: W holds N entries; 0: one entry reused */
    /* LQ family */
    const double *A;          /* n*n   */
    const double *B;          /* n*m   */
    const double *Q;          /* n*n   [*N] : c_xx                     */
    const double *R;          /* m*m   [*N] : c_uu                     */
    const double *P;          /* m*n   [*N] : c_ux                     */
    const double *qv;         /* n     [*N] : linear term in x         */
    const double *rv;         /* m     [*N] : linear term in u         */
    const double *q0;         /* 1     [*N] : constant term            */
    const double *Qf;         /* n*n : h_xx */
    const double *qvf;        /* n   */
    double q0f;
    double kappa;             /* cubic drift coefficient */
    /* power-law family */
    double pl_a, pl_b, pl_p, pl_pu, pl_cx, pl_cu, pl_h;
    /* noise */
    const double *W;          /* n*n [*N] covariance */
} orc_problem;

typedef struct orc_opts {     /* ILEQGSolver keyword arguments, ileqg.jl:191-194 */
    double mu_min, delta_0, lambda, d;
    int64_t iter_max;
    double eps_init, eps_min;
    int32_t adaptive_eps_init;
} orc_opts;

void orc_default_opts(orc_opts *o);

/* ---- a3/a4: rollouts (ileqg.jl:18-38, 62-87) ---------------------------------------- */
int orc_simulate_open(const orc_problem *p, const double *x0, const double *u, double *x);
int orc_simulate_feedback(const orc_problem *p, const double *xbar, const double *l, const double *L,
                          double *x_new, double *u_new);
/* integrate_cost (ileqg.jl:115-124) */
int orc_integrate_cost(const orc_problem *p, const double *x, const double *u, double *cost);
/* Monte-Carlo rollouts under process noise (ileqg.jl:44-55, :94-109); z [K][N][n] injected N(0,1) draws; L == NULL: open loop */
int orc_simulate_noisy(const orc_problem *p, const double *x_nom, const double *l, const double *L, int64_t K,
                       const double *z, double *x_out, double *u_out, double *cost_out);

/* ---- a6: approximate_model (ileqg.jl:258-322) ---------------------------------------
 * outputs (col-major, time slowest): q[N+1], qv[n*(N+1)], Q[n*n*(N+1)], r[m*N], R[m*m*N],
 * P[m*n*N], A[n*n*N], B[n*m*N], W[n*n*N]. */
typedef struct orc_approx {
    double *q, *qv, *Q, *r, *R, *P, *A, *B, *W;
} orc_approx;
orc_approx *orc_approx_alloc(int n, int m, int N);
void orc_approx_free(orc_approx *a);
int orc_approximate_model(const orc_problem *p, const double *u, const double *x, orc_approx *out);

/* ---- a7/a8: risk-sensitive Riccati sweeps (ileqg.jl:341-406, 412-465) ---------------
 * DP result dumps (any may be NULL): s[N+1], sv[n*(N+1)], S[n*n*(N+1)], g[m*N], G[m*n*N], H[m*m*N]. */
typedef struct orc_dp {
    double *s, *sv, *S, *g, *G, *H;
} orc_dp;
orc_dp *orc_dp_alloc(int n, int m, int N);
void orc_dp_free(orc_dp *d);
/* gain sweep: in/out mu, delta (solver state); out L[m*n*N], dl[m*N]. */
int orc_dp_gain(int n, int m, int N, const orc_approx *a, double theta, double mu_min, double delta_0,
                double *mu, double *delta, double *L, double *dl, orc_dp *out);
/* policy evaluation: dl may be NULL (-> zeros). */
int orc_dp_eval(int n, int m, int N, const orc_approx *a, const double *L, const double *dl,
                double theta, double mu, orc_dp *out);

/* ---- a2/a5/a9-a12: the solver object ------------------------------------------------ */
typedef struct orc_solver {
    orc_opts o;
    double mu, delta;
    double eps_init_cur;
    double value_current, d_current;
    int64_t iter_current;
    int n, m, N;
    double *x, *l, *L;          /* nominal trajectory / policy */
    double *eps_hist;           /* pairs (eps, new-current) */
    int64_t n_hist, cap_hist;
    int64_t n_ls_evals;         /* line-search candidates evaluated (incl. DP-failed) */
    orc_approx *ap, *ap_new;
    orc_dp *dp;
    double *dl, *x_new, *u_new, *l_new;
} orc_solver;

orc_solver *orc_solver_new(const orc_problem *p, const orc_opts *o);
void orc_solver_free(orc_solver *s);
void orc_increase_mu_delta(orc_solver *s);       /* ileqg.jl:471-474 */
void orc_decrease_mu_delta(orc_solver *s);       /* ileqg.jl:480-488 */
int orc_initialize(orc_solver *s, const orc_problem *p, const double *x0, const double *u, double theta);
int orc_line_search(orc_solver *s, const orc_problem *p, const double *dl, double theta);
int orc_step(orc_solver *s, const orc_problem *p, double theta);
/* solve! (ileqg.jl:635-659). Returns ORC_OK / ORC_ITER_MAX when a value was produced, an ORC_ERR_* when
 * the reference would have thrown. */
int orc_solve(orc_solver *s, const orc_problem *p, const double *x0, const double *u, double theta);

/* ---- a14/a15: CE cost evaluation (cross_entropy_bilevel_optimization.jl:144-227) ---- */
/* value[i] = solve!(...)[4] or +Inf on exception; cost = value + kl/theta. Optional per-sample
 * status/iters/ls_evals outputs (may be NULL). nthreads>1 uses OpenMP (one sample per thread). */
int orc_compute_value_batch(const orc_problem *p, const orc_opts *o, const double *x0, const double *u,
                            const double *theta, int64_t B, double *value, int32_t *status,
                            int32_t *iters, int32_t *ls_evals, int nthreads);

/* ---- a13/a16-a18: CE solver (cross_entropy_bilevel_optimization.jl:70-138, 233-415) - */
typedef struct orc_ce {
    orc_opts ileqg;
    int64_t num_samples, num_elite, iter_max;
    double lambda;
    int32_t use_theta_max;
    double mu_init, sigma_init, mu, sigma, theta_max, theta_min;
    int64_t iter_current;
    /* injected N(0,1) stream replacing rand(rng, Normal) */
    const double *z; int64_t nz, zpos;
    /* bookkeeping for tests/bench */
    int64_t n_solves, n_redraws;
    int nthreads;
    int64_t n_final_retries;     /* final-solve retries of solve! (:410-413) */
} orc_ce;
void orc_ce_default(orc_ce *c);
void orc_ce_initialize(orc_ce *c);                                   /* :133-138 */
/* get_positive_samples :233-246 ; returns -1 if the z stream ran dry */
int orc_ce_get_positive_samples(orc_ce *c, double mu, double sigma, int64_t num, double *theta);
/* Base.isless on Float64 (the order of sort(by = cost), :326-328): NaN after everything, -0.0 before +0.0 */
int orc_isless(double a, double b);
/* the tail of step! on given thetas / costs: theta_min / theta_max (:314-324), elites, mu, sigma (:326-334) */
void orc_ce_elite_update(orc_ce *c, const double *theta, const double *cost);
/* step! :252-335 ; theta_out/cost_out (size num_samples) receive the last batch; -1 on dry stream */
int orc_ce_step(orc_ce *c, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                double *theta_out, double *cost_out);
/* solve! :364-415 ; outputs x[n*(N+1)], l[m*N], L[m*n*N] (may be NULL) */
int orc_ce_solve(orc_ce *c, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                 double *theta_opt, double *x, double *l, double *L, double *value,
                 double *theta_min, double *theta_max);

/* ---- RAT iLQR++: NelderMeadBilevelOptimizationSolver (nelder_mead_bilevel_optimization.jl:72-352) ---- */
typedef struct orc_nm {
    orc_opts ileqg;
    double alpha, beta, gamma, eps, lambda;
    int64_t iter_max;
    double theta_high_init, theta_low_init;
    int64_t iter_current;
    double theta_high, theta_low;
    int32_t has_c_high, has_c_low;      /* Union{Nothing, Float64}: NOT reset by initialize! (stale across solve! calls) */
    double c_high, c_low;
    int64_t n_solves;
} orc_nm;
void orc_nm_default(orc_nm *s);                                                         /* :102-128 */
void orc_nm_initialize(orc_nm *s);                                                      /* :164-168 */
double orc_nm_compute_cost(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double theta, double kl_bound);  /* :134-158 */
void orc_nm_step(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double kl_bound);   /* :174-252 */
/* solve! :276-352 ; returns the final iLEQG status (a failure there is an uncaught exception in the reference) */
int orc_nm_solve(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                 double *theta_opt, double *x, double *l, double *L, double *value);

/* ---- PETS: CrossEntropyDirectOptimizationSolver (pets.jl) on the generative model family ------------------------
 * Replaces FiniteHorizonGenerativeOptimalControlProblem(f_stochastic, c, h, N) (optimal_control_problems.jl:126-131):
 *   f_stochastic(x, u, rng, use_true_model) = A x + B u + kappa x.^3 + w
 *   w: noise_kind 0 = N(nmean, nchol nchol')   1 = uniform on [nlo, nhi)^n   (test/pets_test.jl:15: rand(rng, n))
 *   use_true_model: 2-component Gaussian mixture (docs example optimal_control_problems.jl:103-110): component 2 with
 *                   probability tw2 has mean tmean2 and Cholesky factor tchol2, otherwise the model noise above.
 *   c(k, x, u) = LQ quadratic form (tables of orc_problem, incl. time variation) + l1u * sum(abs.(u)) ;  h = quadratic.
 * Randomness is injected (serial semantics of compute_cost_serial, pets.jl:128-157): for trajectory j = ii*K + kk and
 * step t the model consumes zn[(j*N + t)*n .. +n) (N(0,1) or U[0,1) draws) and, for the mixture, zu[j*N + t]. */
typedef struct orc_gen_problem {
    orc_problem lq;            /* A, B, kappa, cost tables, Qf ... (W unused) */
    double l1u;
    int32_t noise_kind;
    const double *nmean, *nchol;   /* n, n*n col-major lower */
    double nlo, nhi;
    double tw2;
    const double *tmean2, *tchol2;
} orc_gen_problem;
int orc_pets_compute_cost(const orc_gen_problem *p, const double *x0, const double *controls /* [S][N][m] */,
                          int64_t S, int64_t K, int use_true_model, const double *zn, const double *zu, double *cost);
typedef struct orc_pets {
    int64_t num_control_samples, num_trajectory_samples, num_elite, iter_max;
    double smoothing_factor;
    int64_t N, m, iter_current;
    double *mu_init, *Sigma_init, *mu, *Sigma;     /* [N][m], [N][m*m] col-major; caller-owned */
} orc_pets;
void orc_pets_initialize(orc_pets *s);                                                  /* pets.jl:70-74 */
/* get_elite_samples + compute_new_distribution (pets.jl:159-191): elite_idx[num_elite] out (may be NULL) */
void orc_pets_update(orc_pets *s, const double *controls, const double *cost, int64_t *elite_idx);
/* step! (:193-245), serial semantics; zc = N(0,1) stream for the control samples (S*N*m per step), zn/zu as above */
int orc_pets_step(orc_pets *s, const orc_gen_problem *p, const double *x0, int use_true_model,
                  const double *zc, const double *zn, const double *zu, double *controls_out, double *cost_out);

#ifdef __cplusplus
}
#endif
/* Float64 ^ Float64 as Julia's openlibm computes it (fdlibm e_pow.c restated in fdlibm_pow.h), elementwise */
void orc_pow_array(const double *x, const double *y, long n, double *out);

#endif
