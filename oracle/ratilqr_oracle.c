/*
 * ratilqr_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See ratilqr_oracle.h.
 *
 * Every function cites the reference lines (into /root/reference/src/) it restates.  The
 * operation order of the Riccati step follows the reference as written (left-to-right n-ary `*`
 * of Julia 1.5, D*S formed where the reference forms it, separate factorisations for isposdef,
 * the right-division and logdet), so the timing of this file is also a fair stand-in for the
 * reference's per-solve arithmetic (bench.py cpu_baseline, kind "port").
 */
#include "ratilqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>

#include "fdlibm_pow.h"
#endif

#define MAXD 32                 /* max(n, m) supported by the stack workspaces */
#define IDX(i, j, ld) ((i) + (size_t)(j) * (ld))

#define ORC_ERR_LS_DIVERGED 7

/* ------------------------------------------------------------------------------------------
 * Small dense column-major kernels (stand-ins for Julia's LinearAlgebra / LAPACK calls)
 * ---------------------------------------------------------------------------------------- */

/* C(p x r) = A(p x q) * B(q x r) */
static void mm(int p, int q, int r, const double *A, const double *B, double *C) {
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < p; ++i) {
            double acc = 0.0;
            for (int k = 0; k < q; ++k) acc += A[IDX(i, k, p)] * B[IDX(k, j, q)];
            C[IDX(i, j, p)] = acc;
        }
}
/* C(q x r) = A(p x q)' * B(p x r) */
static void mtm(int p, int q, int r, const double *A, const double *B, double *C) {
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < q; ++i) {
            double acc = 0.0;
            for (int k = 0; k < p; ++k) acc += A[IDX(k, i, p)] * B[IDX(k, j, p)];
            C[IDX(i, j, q)] = acc;
        }
}
static double dot(int n, const double *a, const double *b) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}
/* Symmetric(X): mirror the upper triangle over the lower (Julia's default uplo = :U) */
static void symmetrize_upper(int n, double *X) {
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) X[IDX(i, j, n)] = X[IDX(j, i, n)];
}
/* LU with partial pivoting (getrf). Returns 0, or k+1 if U(k,k) == 0. sign = permutation parity. */
static int lu_factor(int n, double *A, int *piv, int *sign) {
    int info = 0;
    *sign = 1;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[IDX(k, k, n)]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(A[IDX(i, k, n)]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (p != k) {
            *sign = -*sign;
            for (int j = 0; j < n; ++j) {
                double t = A[IDX(k, j, n)]; A[IDX(k, j, n)] = A[IDX(p, j, n)]; A[IDX(p, j, n)] = t;
            }
        }
        double d = A[IDX(k, k, n)];
        if (d == 0.0 || d != d) { if (!info) info = k + 1; continue; }
        for (int i = k + 1; i < n; ++i) A[IDX(i, k, n)] /= d;
        for (int j = k + 1; j < n; ++j) {
            double akj = A[IDX(k, j, n)];
            for (int i = k + 1; i < n; ++i) A[IDX(i, j, n)] -= A[IDX(i, k, n)] * akj;
        }
    }
    return info;
}
static void lu_solve(int n, int r, const double *LU, const int *piv, double *Bm) {
    for (int j = 0; j < r; ++j) {
        double *b = Bm + (size_t)j * n;
        for (int k = 0; k < n; ++k) { int p = piv[k]; if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; } }
        for (int k = 0; k < n; ++k) for (int i = k + 1; i < n; ++i) b[i] -= LU[IDX(i, k, n)] * b[k];
        for (int k = n - 1; k >= 0; --k) {
            b[k] /= LU[IDX(k, k, n)];
            for (int i = 0; i < k; ++i) b[i] -= LU[IDX(i, k, n)] * b[k];
        }
    }
}
/* inv(W) (LU based, as LinearAlgebra.inv). */
static int inv_lu(int n, const double *W, double *Winv) {
    double LU[MAXD * MAXD]; int piv[MAXD], sg;
    memcpy(LU, W, sizeof(double) * n * n);
    if (lu_factor(n, LU, piv, &sg)) return ORC_ERR_SINGULAR;
    memset(Winv, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) Winv[IDX(i, i, n)] = 1.0;
    lu_solve(n, n, LU, piv, Winv);
    return 0;
}
/* isposdef: Cholesky (potrf, upper). U'U = A ; returns 1 if PD and leaves U in Uout (upper). */
static int chol_upper(int n, const double *A, double *U) {
    memset(U, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double d = A[IDX(j, j, n)];
        for (int k = 0; k < j; ++k) d -= U[IDX(k, j, n)] * U[IDX(k, j, n)];
        if (!(d > 0.0) || isinf(d)) return 0;
        double ujj = sqrt(d);
        U[IDX(j, j, n)] = ujj;
        for (int i = j + 1; i < n; ++i) {
            double v = A[IDX(j, i, n)];
            for (int k = 0; k < j; ++k) v -= U[IDX(k, j, n)] * U[IDX(k, i, n)];
            U[IDX(j, i, n)] = v / ujj;
        }
    }
    return 1;
}
/* solve (U'U) X = B in place, B is n x r */
static void chol_solve(int n, int r, const double *U, double *Bm) {
    for (int j = 0; j < r; ++j) {
        double *b = Bm + (size_t)j * n;
        for (int i = 0; i < n; ++i) {            /* U' y = b */
            double v = b[i];
            for (int k = 0; k < i; ++k) v -= U[IDX(k, i, n)] * b[k];
            b[i] = v / U[IDX(i, i, n)];
        }
        for (int i = n - 1; i >= 0; --i) {       /* U x = y */
            double v = b[i];
            for (int k = i + 1; k < n; ++k) v -= U[IDX(i, k, n)] * b[k];
            b[i] = v / U[IDX(i, i, n)];
        }
    }
}
/* logdet(A) of a general matrix via LU (LinearAlgebra.logdet -> logabsdet); DomainError if det < 0 */
static int logdet_lu(int n, const double *A, double *out) {
    double LU[MAXD * MAXD]; int piv[MAXD], sg;
    memcpy(LU, A, sizeof(double) * n * n);
    int info = lu_factor(n, LU, piv, &sg);
    if (info) { *out = -INFINITY; return 0; }
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = LU[IDX(i, i, n)];
        if (d < 0) sg = -sg;
        acc += log(fabs(d));
    }
    if (sg < 0) return ORC_ERR_DOMAIN;
    *out = acc;
    return 0;
}
static double norm2(int n, const double *a) { return sqrt(dot(n, a, a)); }
/* Base.isapprox with default rtol = sqrt(eps(Float64)), atol = 0 */
static int isapprox_default(double x, double y) {
    if (x == y) return 1;
    if (!isfinite(x) || !isfinite(y)) return 0;
    const double rtol = 1.4901161193847656e-8;
    return fabs(x - y) <= rtol * fmax(fabs(x), fabs(y));
}

/* ------------------------------------------------------------------------------------------
 * Model families: f, c, h and their derivatives (stand-in for the user closures + ForwardDiff
 * closures of ileqg.jl:263-273)
 * ---------------------------------------------------------------------------------------- */
static const double *tv(const double *base, int tvflag, int k, size_t sz) { return base + (tvflag ? (size_t)k * sz : 0); }

/* Float64 ^ Float64 of the reference: Julia's openlibm pow (fdlibm), restated in fdlibm_pow.h -- not the host libm's */
static double powchk(double b, double e, int *dom) {
    double r = orc_pow(b, e);
    if (r != r && b == b) *dom = 1;     /* Julia: DomainError for negative base, non-integer exponent */
    return r;
}

static int model_f(const orc_problem *p, const double *x, const double *u, double *xn) {
    int n = p->n, m = p->m;
    if (p->model == ORC_MODEL_LQ) {
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += p->A[IDX(i, j, n)] * x[j];
            double accb = 0.0;
            for (int j = 0; j < m; ++j) accb += p->B[IDX(i, j, n)] * u[j];
            acc += accb;
            if (p->kappa != 0.0) acc += p->kappa * (x[i] * x[i] * x[i]);
            xn[i] = acc;
        }
        return 0;
    } else {
        int dom = 0;
        for (int i = 0; i < n; ++i) xn[i] = powchk(x[i], p->pl_a, &dom) + powchk(u[i], p->pl_b, &dom);
        return dom ? ORC_ERR_DOMAIN : 0;
    }
}
static int model_c(const orc_problem *p, int k, const double *x, const double *u, double *c) {
    int n = p->n, m = p->m;
    if (p->model == ORC_MODEL_LQ) {
        const double *Q = tv(p->Q, p->cost_tv, k, (size_t)n * n), *R = tv(p->R, p->cost_tv, k, (size_t)m * m);
        const double *P = tv(p->P, p->cost_tv, k, (size_t)m * n);
        const double *qv = tv(p->qv, p->cost_tv, k, n), *rv = tv(p->rv, p->cost_tv, k, m);
        double q0 = *tv(p->q0, p->cost_tv, k, 1);
        double xQx = 0, uRu = 0, uPx = 0;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) xQx += x[i] * Q[IDX(i, j, n)] * x[j];
        for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) uRu += u[i] * R[IDX(i, j, m)] * u[j];
        for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) uPx += u[i] * P[IDX(i, j, m)] * x[j];
        *c = 0.5 * xQx + 0.5 * uRu + uPx + dot(n, qv, x) + dot(m, rv, u) + q0;
        return 0;
    } else {
        int dom = 0; double acc = 0;
        for (int i = 0; i < n; ++i) acc += p->pl_cx * powchk(x[i], p->pl_p, &dom) + p->pl_cu * powchk(u[i], p->pl_pu, &dom);
        *c = acc;
        return dom ? ORC_ERR_DOMAIN : 0;
    }
}
static int model_h(const orc_problem *p, const double *x, double *h) {
    int n = p->n;
    if (p->model == ORC_MODEL_LQ) {
        double xQx = 0;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) xQx += x[i] * p->Qf[IDX(i, j, n)] * x[j];
        *h = 0.5 * xQx + dot(n, p->qvf, x) + p->q0f;
    } else {
        *h = p->pl_h;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Rollouts and cost integration
 * ---------------------------------------------------------------------------------------- */
/* simulate_dynamics(problem, x_0, u_array)  -- ileqg.jl:18-38 */
int orc_simulate_open(const orc_problem *p, const double *x0, const double *u, double *x) {
    int n = p->n, m = p->m;
    memcpy(x, x0, sizeof(double) * n);
    for (int t = 0; t < p->N; ++t) {
        int rc = model_f(p, x + (size_t)t * n, u + (size_t)t * m, x + (size_t)(t + 1) * n);
        if (rc) return rc;
    }
    return 0;
}
/* simulate_dynamics(problem, x_array, l_array, L_array)  -- ileqg.jl:62-87 */
int orc_simulate_feedback(const orc_problem *p, const double *xbar, const double *l, const double *L,
                          double *x_new, double *u_new) {
    int n = p->n, m = p->m;
    double dx[MAXD];
    memcpy(x_new, xbar, sizeof(double) * n);
    for (int t = 0; t < p->N; ++t) {
        const double *Lt = L + (size_t)t * m * n;
        for (int j = 0; j < n; ++j) dx[j] = x_new[(size_t)t * n + j] - xbar[(size_t)t * n + j];
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += Lt[IDX(i, j, m)] * dx[j];
            u_new[(size_t)t * m + i] = l[(size_t)t * m + i] + acc;
        }
        int rc = model_f(p, x_new + (size_t)t * n, u_new + (size_t)t * m, x_new + (size_t)(t + 1) * n);
        if (rc) return rc;
    }
    return 0;
}
static int chol_lower(int n, const double *A, double *Lo);
/* simulate_dynamics with process noise -- ileqg.jl:44-55 (open loop, L == NULL) and :94-109 (affine feedback policy):
 * K independent rollouts, x_{k+1} = f(x_k, u_k) + w_k, w_k ~ N(0, W(k)) drawn as chol_lower(W(k)) z_k from the injected
 * standard-normal stream z [K][N][n] (Distributions.jl MvNormal sampling = unwhitening by the lower Cholesky factor; the
 * reference's RNG stream itself is not reproducible: parity unpinned there).  x_nom is the nominal state array (only its first
 * column is used in the open-loop form).  cost_out[k] = integrate_cost of rollout k (ileqg.jl:115-124).  Outputs may be NULL. */
int orc_simulate_noisy(const orc_problem *p, const double *x_nom, const double *l, const double *L, int64_t K,
                       const double *z, double *x_out, double *u_out, double *cost_out) {
    const int n = p->n, m = p->m, N = p->N;
    double *x = (double *)malloc(sizeof(double) * (size_t)n * (N + 1)), *u = (double *)malloc(sizeof(double) * (size_t)m * N);
    double *Lc = (double *)malloc(sizeof(double) * (size_t)n * n * N), dx[MAXD];
    int rc = 0;
    for (int t = 0; t < N && !rc; ++t)
        if (!chol_lower(n, tv(p->W, p->W_tv, t, (size_t)n * n), Lc + (size_t)t * n * n)) rc = ORC_ERR_DOMAIN;
    for (int64_t k = 0; k < K && !rc; ++k) {
        memcpy(x, x_nom, sizeof(double) * n);
        for (int t = 0; t < N && !rc; ++t) {
            const double *xt = x + (size_t)t * n;
            double *ut = u + (size_t)t * m, *xn = x + (size_t)(t + 1) * n;
            for (int i = 0; i < m; ++i) ut[i] = l[(size_t)t * m + i];
            if (L) {
                const double *Lt = L + (size_t)t * m * n;
                for (int j = 0; j < n; ++j) dx[j] = xt[j] - x_nom[(size_t)t * n + j];
                for (int i = 0; i < m; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < n; ++j) acc += Lt[IDX(i, j, m)] * dx[j];
                    ut[i] = l[(size_t)t * m + i] + acc;
                }
            }
            rc = model_f(p, xt, ut, xn);
            const double *zt = z + ((size_t)k * N + t) * n, *Lw = Lc + (size_t)t * n * n;
            for (int i = 0; i < n; ++i) {
                double w = 0.0;
                for (int j = 0; j <= i; ++j) w += Lw[IDX(i, j, n)] * zt[j];
                xn[i] += w;
            }
        }
        if (rc) break;
        if (x_out) memcpy(x_out + (size_t)k * n * (N + 1), x, sizeof(double) * (size_t)n * (N + 1));
        if (u_out) memcpy(u_out + (size_t)k * m * N, u, sizeof(double) * (size_t)m * N);
        if (cost_out) rc = orc_integrate_cost(p, x, u, &cost_out[k]);
    }
    free(x); free(u); free(Lc);
    return rc;
}
/* integrate_cost -- ileqg.jl:115-124 */
int orc_integrate_cost(const orc_problem *p, const double *x, const double *u, double *cost) {
    double acc = 0.0, c;
    for (int t = 0; t < p->N; ++t) {
        int rc = model_c(p, t, x + (size_t)t * p->n, u + (size_t)t * p->m, &c);
        if (rc) return rc;
        acc += c;
    }
    model_h(p, x + (size_t)p->N * p->n, &c);
    *cost = acc + c;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * approximate_model -- ileqg.jl:258-322
 * ---------------------------------------------------------------------------------------- */
orc_approx *orc_approx_alloc(int n, int m, int N) {
    orc_approx *a = (orc_approx *)calloc(1, sizeof(*a));
    a->q = (double *)calloc((size_t)N + 1, sizeof(double));
    a->qv = (double *)calloc((size_t)n * (N + 1), sizeof(double));
    a->Q = (double *)calloc((size_t)n * n * (N + 1), sizeof(double));
    a->r = (double *)calloc((size_t)m * N, sizeof(double));
    a->R = (double *)calloc((size_t)m * m * N, sizeof(double));
    a->P = (double *)calloc((size_t)m * n * N, sizeof(double));
    a->A = (double *)calloc((size_t)n * n * N, sizeof(double));
    a->B = (double *)calloc((size_t)n * m * N, sizeof(double));
    a->W = (double *)calloc((size_t)n * n * N, sizeof(double));
    return a;
}
void orc_approx_free(orc_approx *a) {
    if (!a) return;
    free(a->q); free(a->qv); free(a->Q); free(a->r); free(a->R); free(a->P); free(a->A); free(a->B); free(a->W);
    free(a);
}

int orc_approximate_model(const orc_problem *p, const double *u_arr, const double *x_arr, orc_approx *o) {
    int n = p->n, m = p->m, N = p->N, dom = 0;
    for (int t = 0; t < N; ++t) {                                   /* ileqg.jl:294-313 */
        const double *x = x_arr + (size_t)t * n, *u = u_arr + (size_t)t * m;
        double *q = o->q + t, *qv = o->qv + (size_t)t * n, *Q = o->Q + (size_t)t * n * n;
        double *r = o->r + (size_t)t * m, *R = o->R + (size_t)t * m * m, *P = o->P + (size_t)t * m * n;
        double *A = o->A + (size_t)t * n * n, *B = o->B + (size_t)t * n * m, *W = o->W + (size_t)t * n * n;
        int rc = model_c(p, t, x, u, q);                             /* :296 */
        if (rc) return rc;
        if (p->model == ORC_MODEL_LQ) {
            const double *Qk = tv(p->Q, p->cost_tv, t, (size_t)n * n), *Rk = tv(p->R, p->cost_tv, t, (size_t)m * m);
            const double *Pk = tv(p->P, p->cost_tv, t, (size_t)m * n);
            const double *qvk = tv(p->qv, p->cost_tv, t, n), *rvk = tv(p->rv, p->cost_tv, t, m);
            for (int i = 0; i < n; ++i) {                            /* cx :297 */
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += Qk[IDX(i, j, n)] * x[j];
                double accp = 0.0;
                for (int j = 0; j < m; ++j) accp += Pk[IDX(j, i, m)] * u[j];
                qv[i] = acc + accp + qvk[i];
            }
            memcpy(Q, Qk, sizeof(double) * n * n); symmetrize_upper(n, Q);      /* cxx :298 */
            for (int i = 0; i < m; ++i) {                            /* cu :299 */
                double acc = 0.0;
                for (int j = 0; j < m; ++j) acc += Rk[IDX(i, j, m)] * u[j];
                double accp = 0.0;
                for (int j = 0; j < n; ++j) accp += Pk[IDX(i, j, m)] * x[j];
                r[i] = acc + accp + rvk[i];
            }
            memcpy(R, Rk, sizeof(double) * m * m); symmetrize_upper(m, R);      /* cuu :300 */
            memcpy(P, Pk, sizeof(double) * m * n);                              /* cux :301 */
            memcpy(A, p->A, sizeof(double) * n * n);                            /* fx  :303 */
            if (p->kappa != 0.0)
                for (int i = 0; i < n; ++i) A[IDX(i, i, n)] += 3.0 * p->kappa * (x[i] * x[i]);
            memcpy(B, p->B, sizeof(double) * n * m);                            /* fu  :308 */
        } else {
            memset(Q, 0, sizeof(double) * n * n); memset(R, 0, sizeof(double) * m * m);
            memset(P, 0, sizeof(double) * m * n); memset(A, 0, sizeof(double) * n * n);
            memset(B, 0, sizeof(double) * n * m);
            for (int i = 0; i < n; ++i) {
                qv[i] = p->pl_cx * p->pl_p * powchk(x[i], p->pl_p - 1.0, &dom);
                Q[IDX(i, i, n)] = p->pl_cx * p->pl_p * (p->pl_p - 1.0) * powchk(x[i], p->pl_p - 2.0, &dom);
                A[IDX(i, i, n)] = p->pl_a * powchk(x[i], p->pl_a - 1.0, &dom);
            }
            for (int i = 0; i < m; ++i) {
                r[i] = p->pl_cu * p->pl_pu * powchk(u[i], p->pl_pu - 1.0, &dom);
                R[IDX(i, i, m)] = p->pl_cu * p->pl_pu * (p->pl_pu - 1.0) * powchk(u[i], p->pl_pu - 2.0, &dom);
                B[IDX(i, i, n)] = p->pl_b * powchk(u[i], p->pl_b - 1.0, &dom);
            }
        }
        memcpy(W, tv(p->W, p->W_tv, t, (size_t)n * n), sizeof(double) * n * n);  /* :312 */
    }
    {                                                                /* terminal :314-316 */
        const double *x = x_arr + (size_t)N * n;
        double *qv = o->qv + (size_t)N * n, *Q = o->Q + (size_t)N * n * n;
        model_h(p, x, o->q + N);
        if (p->model == ORC_MODEL_LQ) {
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += p->Qf[IDX(i, j, n)] * x[j];
                qv[i] = acc + p->qvf[i];
            }
            memcpy(Q, p->Qf, sizeof(double) * n * n); symmetrize_upper(n, Q);
        } else {
            memset(qv, 0, sizeof(double) * n); memset(Q, 0, sizeof(double) * n * n);
        }
    }
    return dom ? ORC_ERR_DOMAIN : 0;
}

/* ------------------------------------------------------------------------------------------
 * Riccati-like sweeps
 * ---------------------------------------------------------------------------------------- */
orc_dp *orc_dp_alloc(int n, int m, int N) {
    orc_dp *d = (orc_dp *)calloc(1, sizeof(*d));
    d->s = (double *)calloc((size_t)N + 1, sizeof(double));
    d->sv = (double *)calloc((size_t)n * (N + 1), sizeof(double));
    d->S = (double *)calloc((size_t)n * n * (N + 1), sizeof(double));
    d->g = (double *)calloc((size_t)m * N, sizeof(double));
    d->G = (double *)calloc((size_t)m * n * N, sizeof(double));
    d->H = (double *)calloc((size_t)m * m * N, sizeof(double));
    return d;
}
void orc_dp_free(orc_dp *d) {
    if (!d) return;
    free(d->s); free(d->sv); free(d->S); free(d->g); free(d->G); free(d->H); free(d);
}

/* One backward step, common to ileqg.jl:361-391 (gain = 1) and :435-460 (gain = 0).
 * Return: 0 ok, ORC_ERR_M_NOT_PD_* (caller maps), -1 = H not PD (gain sweep only), other ORC_ERR_*. */
static int dp_step(int n, int m, int gain, double theta, double mu,
                   double q, const double *qv, const double *Q, const double *r, const double *R, const double *P,
                   const double *A, const double *B, const double *W,
                   double s1, const double *sv1, const double *S1,
                   double *L /* in (eval) / out (gain) */, double *dl /* in/out */,
                   double *s0, double *sv0, double *S0, double *g, double *G, double *H) {
    double Winv[MAXD * MAXD], M[MAXD * MAXD], U[MAXD * MAXD], D[MAXD * MAXD], DS[MAXD * MAXD];
    double t1[MAXD * MAXD], t2[MAXD * MAXD], t3[MAXD * MAXD], v1[MAXD], v2[MAXD];
    int rc;
    if ((rc = inv_lu(n, W, Winv))) return rc;                                   /* inv(W) :365 */
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) M[IDX(i, j, n)] = Winv[IDX(i, j, n)] - theta * S1[IDX(i, j, n)];
    symmetrize_upper(n, M);                                                     /* Symmetric(...) */
    if (!chol_upper(n, M, U)) return ORC_ERR_M_NOT_PD_GAIN;                      /* @assert isposdef(M) :366 */
    /* D = I + theta.*S/M  (:367).  X = (theta S) M^-1  <=>  M X' = (theta S)' */
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) t1[IDX(i, j, n)] = theta * S1[IDX(j, i, n)];
    chol_solve(n, n, U, t1);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) D[IDX(i, j, n)] = (i == j ? 1.0 : 0.0) + t1[IDX(j, i, n)];
    /* g = r + B'*D*s_vec  (:368), evaluated as (B'*D)*s_vec */
    mtm(n, m, n, B, D, t1);                    /* t1 = B'D (m x n) */
    for (int i = 0; i < m; ++i) { double acc = 0; for (int k = 0; k < n; ++k) acc += t1[IDX(i, k, m)] * sv1[k]; g[i] = r[i] + acc; }
    /* G = P + B'*(D*S)*A  (:369) */
    mm(n, n, n, D, S1, DS);
    mtm(n, m, n, B, DS, t1);                   /* B'(DS) (m x n) */
    mm(m, n, n, t1, A, t2);
    for (int i = 0; i < m * n; ++i) G[i] = P[i] + t2[i];
    /* H = R + B'*(D*S)*B + mu*I ; Symmetric (:370-371) */
    mm(m, n, m, t1, B, t2);
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) H[IDX(i, j, m)] = R[IDX(i, j, m)] + t2[IDX(i, j, m)] + (i == j ? mu : 0.0);
    symmetrize_upper(m, H);
    if (gain) {
        if (!chol_upper(m, H, t2)) return -1;                                   /* !isposdef(H) :372 */
        double nH[MAXD * MAXD]; int piv[MAXD], sg;
        for (int i = 0; i < m * m; ++i) nH[i] = -H[i];                           /* (-H)\G  :379 */
        if (lu_factor(m, nH, piv, &sg)) return ORC_ERR_SINGULAR;
        memcpy(L, G, sizeof(double) * m * n); lu_solve(m, n, nH, piv, L);
        memcpy(dl, g, sizeof(double) * m);    lu_solve(m, 1, nH, piv, dl);       /* :381 */
    }
    /* s = q + s1 + 0.5*dl'*H*dl + dl'*g  (:383) */
    for (int j = 0; j < m; ++j) { double acc = 0; for (int i = 0; i < m; ++i) acc += (0.5 * dl[i]) * H[IDX(i, j, m)]; v1[j] = acc; }
    double sval = q + s1 + dot(m, v1, dl) + dot(m, dl, g);
    if (theta == 0.0) {                                                         /* :384-385 */
        mm(n, n, n, W, S1, t2);
        double tr = 0; for (int i = 0; i < n; ++i) tr += t2[IDX(i, i, n)];
        sval += 0.5 * tr;
    } else {                                                                    /* :387 */
        for (int i = 0; i < n; ++i) v1[i] = theta / 2 * sv1[i];
        chol_solve(n, 1, U, v1);                     /* (theta/2 s_vec')/M  -> M \ (theta/2 s_vec) */
        mm(n, n, n, W, M, t2);
        double ld;
        if ((rc = logdet_lu(n, t2, &ld))) return rc;
        sval += dot(n, v1, sv1) - 1 / (2 * theta) * ld;
    }
    *s0 = sval;
    /* s_vec = q_vec + A'*D*s_vec1 + L'*H*dl + L'*g + G'*dl  (:389) */
    mtm(n, n, n, A, D, t2);                       /* A'D */
    for (int i = 0; i < n; ++i) { double acc = 0; for (int k = 0; k < n; ++k) acc += t2[IDX(i, k, n)] * sv1[k]; v1[i] = acc; }
    mtm(m, n, m, L, H, t3);                       /* L'H (n x m) */
    for (int i = 0; i < n; ++i) {
        double a1 = 0, a2 = 0, a3 = 0;
        for (int k = 0; k < m; ++k) { a1 += t3[IDX(i, k, n)] * dl[k]; a2 += L[IDX(k, i, m)] * g[k]; a3 += G[IDX(k, i, m)] * dl[k]; }
        v2[i] = qv[i] + v1[i] + a1 + a2 + a3;
    }
    /* S = Q + A'*D*S1*A + L'*H*L + L'*G + G'*L ; Symmetric  (:390-391) */
    double t4[MAXD * MAXD], t5[MAXD * MAXD];
    mm(n, n, n, t2, S1, t4);                      /* (A'D) S1 */
    mm(n, n, n, t4, A, t5);                       /* ... A */
    mm(n, m, n, t3, L, t4);                       /* (L'H) L */
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        double lg = 0, gl = 0;
        for (int k = 0; k < m; ++k) { lg += L[IDX(k, i, m)] * G[IDX(k, j, m)]; gl += G[IDX(k, i, m)] * L[IDX(k, j, m)]; }
        S0[IDX(i, j, n)] = Q[IDX(i, j, n)] + t5[IDX(i, j, n)] + t4[IDX(i, j, n)] + lg + gl;
    }
    symmetrize_upper(n, S0);
    memcpy(sv0, v2, sizeof(double) * n);
    return 0;
}

/* solve_approximate_dp!  -- ileqg.jl:341-406 */
int orc_dp_gain(int n, int m, int N, const orc_approx *a, double theta, double mu_min, double delta_0,
                double *mu, double *delta, double *L, double *dl, orc_dp *out) {
    orc_dp *d = out ? out : orc_dp_alloc(n, m, N);
    int rc = 0;
    d->s[N] = a->q[N];                                                          /* :352-354 */
    memcpy(d->sv + (size_t)N * n, a->qv + (size_t)N * n, sizeof(double) * n);
    memcpy(d->S + (size_t)N * n * n, a->Q + (size_t)N * n * n, sizeof(double) * n * n);
    symmetrize_upper(n, d->S + (size_t)N * n * n);
    int all_psd = 0, restarts = 0;
    while (!all_psd) {                                                          /* :359 */
        int broke = 0;
        for (int t = N - 1; t >= 0; --t) {
            rc = dp_step(n, m, 1, theta, *mu, a->q[t], a->qv + (size_t)t * n, a->Q + (size_t)t * n * n,
                         a->r + (size_t)t * m, a->R + (size_t)t * m * m, a->P + (size_t)t * m * n,
                         a->A + (size_t)t * n * n, a->B + (size_t)t * n * m, a->W + (size_t)t * n * n,
                         d->s[t + 1], d->sv + (size_t)(t + 1) * n, d->S + (size_t)(t + 1) * n * n,
                         L + (size_t)t * m * n, dl + (size_t)t * m,
                         d->s + t, d->sv + (size_t)t * n, d->S + (size_t)t * n * n,
                         d->g + (size_t)t * m, d->G + (size_t)t * m * n, d->H + (size_t)t * m * m);
            if (rc == -1) {                                                     /* :372-378 increase_mu_and_delta!, break */
                *delta = fmax(delta_0, *delta * delta_0);
                *mu = fmax(mu_min, *mu * *delta);
                broke = 1; rc = 0;
                break;
            }
            if (rc) goto done;
        }
        if (!broke) all_psd = 1;
        else if (++restarts > 400 || !isfinite(*mu)) { rc = ORC_ERR_MU_DIVERGED; goto done; }
    }
done:
    if (!out) orc_dp_free(d);
    return rc;
}

/* solve_approximate_dp  -- ileqg.jl:412-465 */
int orc_dp_eval(int n, int m, int N, const orc_approx *a, const double *L, const double *dl,
                double theta, double mu, orc_dp *out) {
    orc_dp *d = out ? out : orc_dp_alloc(n, m, N);
    int rc = 0;
    double zero[MAXD] = {0}, dlt[MAXD], Lt[MAXD * MAXD];
    d->s[N] = a->q[N];                                                          /* :429-431 */
    memcpy(d->sv + (size_t)N * n, a->qv + (size_t)N * n, sizeof(double) * n);
    memcpy(d->S + (size_t)N * n * n, a->Q + (size_t)N * n * n, sizeof(double) * n * n);
    symmetrize_upper(n, d->S + (size_t)N * n * n);
    for (int t = N - 1; t >= 0; --t) {                                          /* :434 */
        memcpy(dlt, dl ? dl + (size_t)t * m : zero, sizeof(double) * m);         /* :447-451 */
        memcpy(Lt, L + (size_t)t * m * n, sizeof(double) * m * n);
        rc = dp_step(n, m, 0, theta, mu, a->q[t], a->qv + (size_t)t * n, a->Q + (size_t)t * n * n,
                     a->r + (size_t)t * m, a->R + (size_t)t * m * m, a->P + (size_t)t * m * n,
                     a->A + (size_t)t * n * n, a->B + (size_t)t * n * m, a->W + (size_t)t * n * n,
                     d->s[t + 1], d->sv + (size_t)(t + 1) * n, d->S + (size_t)(t + 1) * n * n,
                     Lt, dlt, d->s + t, d->sv + (size_t)t * n, d->S + (size_t)t * n * n,
                     d->g + (size_t)t * m, d->G + (size_t)t * m * n, d->H + (size_t)t * m * m);
        if (rc) break;
    }
    if (!out) orc_dp_free(d);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * ILEQGSolver
 * ---------------------------------------------------------------------------------------- */
void orc_default_opts(orc_opts *o) {                                            /* ileqg.jl:191-194 */
    o->mu_min = 1e-6; o->delta_0 = 2.0; o->lambda = 0.5; o->d = 1e-2; o->iter_max = 100;
    o->eps_init = 1.0; o->eps_min = 1e-6; o->adaptive_eps_init = 0;
}

orc_solver *orc_solver_new(const orc_problem *p, const orc_opts *o) {          /* ileqg.jl:191-208 */
    if (p->n > MAXD || p->m > MAXD) return NULL;
    if (!(0 < o->lambda && o->lambda < 1) || !(o->d > 0) || !(o->mu_min > 0) || !(o->delta_0 > 0) ||
        !(0 < o->eps_init && o->eps_init <= 1) || !(o->eps_init > o->eps_min) || !(0 < o->eps_min && o->eps_min < 1))
        return NULL;                                                            /* the @assert block :195-201 */
    orc_solver *s = (orc_solver *)calloc(1, sizeof(*s));
    int n = p->n, m = p->m, N = p->N;
    s->o = *o; s->n = n; s->m = m; s->N = N;
    s->mu = o->mu_min; s->delta = o->delta_0;                                   /* :206 */
    s->eps_init_cur = o->eps_init;
    s->value_current = INFINITY; s->d_current = INFINITY; s->iter_current = 0;
    s->x = (double *)calloc((size_t)n * (N + 1), sizeof(double));
    s->l = (double *)calloc((size_t)m * N, sizeof(double));
    s->L = (double *)calloc((size_t)m * n * N, sizeof(double));
    s->x_new = (double *)calloc((size_t)n * (N + 1), sizeof(double));
    s->u_new = (double *)calloc((size_t)m * N, sizeof(double));
    s->l_new = (double *)calloc((size_t)m * N, sizeof(double));
    s->dl = (double *)calloc((size_t)m * N, sizeof(double));
    s->ap = orc_approx_alloc(n, m, N); s->ap_new = orc_approx_alloc(n, m, N);
    s->dp = orc_dp_alloc(n, m, N);
    s->cap_hist = 64; s->eps_hist = (double *)calloc(2 * (size_t)s->cap_hist, sizeof(double));
    return s;
}
void orc_solver_free(orc_solver *s) {
    if (!s) return;
    free(s->x); free(s->l); free(s->L); free(s->x_new); free(s->u_new); free(s->l_new); free(s->dl);
    orc_approx_free(s->ap); orc_approx_free(s->ap_new); orc_dp_free(s->dp); free(s->eps_hist); free(s);
}
void orc_increase_mu_delta(orc_solver *s) {                                     /* ileqg.jl:471-474 */
    s->delta = fmax(s->o.delta_0, s->delta * s->o.delta_0);
    s->mu = fmax(s->o.mu_min, s->mu * s->delta);
}
void orc_decrease_mu_delta(orc_solver *s) {                                     /* ileqg.jl:480-488 */
    s->delta = fmin(1 / s->o.delta_0, s->delta / s->o.delta_0);
    double cand = s->mu * s->delta;
    s->mu = (cand >= s->o.mu_min) ? cand : 0.0;
}

/* initialize!  -- ileqg.jl:214-236 */
int orc_initialize(orc_solver *s, const orc_problem *p, const double *x0, const double *u, double theta) {
    int n = s->n, m = s->m, N = s->N, rc;
    s->mu = 0.0; s->delta = s->o.delta_0;                                       /* :216 */
    s->d_current = INFINITY; s->iter_current = 0;
    s->eps_init_cur = s->o.eps_init; s->n_hist = 0; s->n_ls_evals = 0;
    if ((rc = orc_simulate_open(p, x0, u, s->x))) return rc;                    /* :225 */
    memcpy(s->l, u, sizeof(double) * m * N);                                    /* :228 */
    memset(s->L, 0, sizeof(double) * m * n * N);                                /* :230-232 */
    if ((rc = orc_approximate_model(p, s->l, s->x, s->ap))) return rc;          /* :233 */
    rc = orc_dp_eval(n, m, N, s->ap, s->L, NULL, theta, s->mu, s->dp);          /* :234 (not in a try) */
    if (rc == ORC_ERR_M_NOT_PD_GAIN) return ORC_ERR_M_NOT_PD_INIT;
    if (rc) return rc;
    s->value_current = s->dp->s[0];                                             /* :235 */
    return 0;
}

static void push_hist(orc_solver *s, double eps, double dv) {
    if (s->n_hist == s->cap_hist) {
        s->cap_hist *= 2;
        s->eps_hist = (double *)realloc(s->eps_hist, 2 * (size_t)s->cap_hist * sizeof(double));
    }
    s->eps_hist[2 * s->n_hist] = eps; s->eps_hist[2 * s->n_hist + 1] = dv; s->n_hist++;
}
static double max_step_norm(int m, int N, const double *l, const double *u_new) {  /* maximum(norm.(l .- u_new)) */
    double best = -INFINITY, diff[MAXD];
    for (int t = 0; t < N; ++t) {
        for (int i = 0; i < m; ++i) diff[i] = l[(size_t)t * m + i] - u_new[(size_t)t * m + i];
        double v = norm2(m, diff);
        if (v > best || v != v) best = v;       /* Julia's maximum propagates NaN */
        if (v != v) break;
    }
    return best;
}

/* line_search!  -- ileqg.jl:494-592 */
int orc_line_search(orc_solver *s, const orc_problem *p, const double *dl, double theta) {
    int n = s->n, m = s->m, N = s->N, rc;
    double cur = s->value_current;                                              /* :497 */
    double eps = s->eps_init_cur;                                               /* :502 */
    int64_t count = 0;
    for (;;) {
        count++;                                                                /* :505 */
        if (count > 4000) return ORC_ERR_LS_DIVERGED;       /* reference would spin (App. B.5) */
        s->n_ls_evals++;
        for (int i = 0; i < m * N; ++i) s->l_new[i] = s->l[i] + eps * dl[i];     /* :509 */
        if ((rc = orc_simulate_feedback(p, s->x, s->l_new, s->L, s->x_new, s->u_new))) return rc;  /* :517 */
        if ((rc = orc_approximate_model(p, s->u_new, s->x_new, s->ap_new))) return rc;             /* :520 */
        rc = orc_dp_eval(n, m, N, s->ap_new, s->L, NULL, theta, s->mu, s->dp);  /* :522-528 try/catch */
        if (rc) { eps *= s->o.lambda; continue; }                               /* :529-535 */
        double newv = s->dp->s[0];                                              /* :536 */
        push_hist(s, eps, newv - cur);                                          /* :537 */
        int accept = isapprox_default(newv, cur) || newv < cur;                 /* :538 */
        if (!accept) {
            eps *= s->o.lambda;                                                 /* :557 */
            if (!(eps < s->o.eps_min)) continue;                                /* :558 */
        }
        s->d_current = max_step_norm(m, N, s->l, s->u_new);                     /* :539 / :559 */
        s->value_current = newv;
        memcpy(s->x, s->x_new, sizeof(double) * n * (N + 1));
        memcpy(s->l, s->u_new, sizeof(double) * m * N);
        break;
    }
    if (s->o.adaptive_eps_init) {                                               /* :582-591 */
        if (count == 1) s->eps_init_cur = fmin(s->o.eps_init, eps / s->o.lambda);
        else { while (eps < s->o.eps_min) eps = eps / s->o.lambda; s->eps_init_cur = eps; }
    }
    return 0;
}

/* step!  -- ileqg.jl:598-613 */
int orc_step(orc_solver *s, const orc_problem *p, double theta) {
    int rc;
    s->iter_current++;
    if ((rc = orc_approximate_model(p, s->l, s->x, s->ap))) return rc;          /* :604 */
    rc = orc_dp_gain(s->n, s->m, s->N, s->ap, theta, s->o.mu_min, s->o.delta_0, &s->mu, &s->delta,
                     s->L, s->dl, s->dp);                                       /* :610-611 */
    if (rc) return rc;
    return orc_line_search(s, p, s->dl, theta);                                 /* :612 */
}

/* solve!  -- ileqg.jl:635-659 */
int orc_solve(orc_solver *s, const orc_problem *p, const double *x0, const double *u, double theta) {
    int rc = orc_initialize(s, p, x0, u, theta);                                /* :639 */
    if (rc) return rc;
    for (;;) {
        if ((rc = orc_step(s, p, theta))) return rc;                            /* :641 */
        if (s->o.d > s->d_current && s->mu <= s->o.mu_min) return ORC_OK;       /* :642 */
        else if (s->iter_current == s->o.iter_max) return ORC_ITER_MAX;         /* :648 */
    }
}

/* compute_value_worker / compute_cost_serial  -- cross_entropy_bilevel_optimization.jl:144-167, 198-227 */
int orc_compute_value_batch(const orc_problem *p, const orc_opts *o, const double *x0, const double *u,
                            const double *theta, int64_t B, double *value, int32_t *status,
                            int32_t *iters, int32_t *ls_evals, int nthreads) {
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
    /* One solver object per worker thread, re-initialised by initialize! for every sample: numerically identical to the
     * reference's fresh ILEQGSolver per sample (:148; initialize! resets every field a solve reads), but without 256
     * threads contending in malloc -- so that the reported CPU baseline is not an allocator benchmark. */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_solver *s = orc_solver_new(p, o);
        if (!s) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            bad = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int64_t i = 0; i < B; ++i) {
                s->mu = o->mu_min; s->delta = o->delta_0;                           /* constructor state (:206) */
                int rc = orc_solve(s, p, x0, u, theta[i]);
                value[i] = (rc == ORC_OK || rc == ORC_ITER_MAX) ? s->value_current : INFINITY;   /* catch -> Inf :163 */
                if (status) status[i] = rc;
                if (iters) iters[i] = (int32_t)s->iter_current;
                if (ls_evals) ls_evals[i] = (int32_t)s->n_ls_evals;
            }
            orc_solver_free(s);
        }
    }
    (void)nthreads;
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * CrossEntropyBilevelOptimizationSolver
 * ---------------------------------------------------------------------------------------- */
void orc_ce_default(orc_ce *c) {                                                /* :100-127 */
    memset(c, 0, sizeof(*c));
    orc_default_opts(&c->ileqg);
    c->mu_init = 1.0; c->sigma_init = 2.0; c->num_samples = 10; c->num_elite = 3; c->iter_max = 5;
    c->lambda = 0.5; c->use_theta_max = 0;
    c->mu = c->mu_init; c->sigma = c->sigma_init; c->theta_max = 0.0; c->theta_min = INFINITY;
    c->nthreads = 1;
}
void orc_ce_initialize(orc_ce *c) {                                             /* :133-138 */
    c->iter_current = 0; c->mu = c->mu_init; c->sigma = c->sigma_init;
    c->theta_max = 0.0; c->theta_min = INFINITY;
}
int orc_ce_get_positive_samples(orc_ce *c, double mu, double sigma, int64_t num, double *theta) {  /* :233-246 */
    int64_t k = 0;
    for (;;) {
        if (c->zpos >= c->nz) return -1;
        double th = mu + sigma * c->z[c->zpos++];       /* rand(rng, Normal(mu, sigma)) */
        if (th > 0.0) theta[k++] = th;
        if (k >= num) break;
    }
    return 0;
}

typedef struct { double theta, cost; int64_t idx; } pair_t;
static int pair_less(double a, double b) {            /* Base.isless on Float64: NaN sorts last, -0.0 before +0.0 */
    if (a != a) return 0;
    if (b != b) return 1;
    return a < b || (a == b && signbit(a) && !signbit(b));
}
static void stable_sort_pairs(pair_t *v, int64_t n) { /* insertion sort: stable, like sort(by=...) */
    for (int64_t i = 1; i < n; ++i) {
        pair_t key = v[i]; int64_t j = i - 1;
        while (j >= 0 && pair_less(key.cost, v[j].cost)) { v[j + 1] = v[j]; --j; }
        v[j + 1] = key;
    }
}

/* tail of step! on given costs -- theta_min / theta_max (:314-324), elites by sort(by = cost) under isless, mean and population std (:326-334) */
int orc_isless(double a, double b) { return pair_less(a, b); }
void orc_ce_elite_update(orc_ce *c, const double *theta, const double *cost) {
    int64_t B = c->num_samples;
    for (int64_t i = 0; i < B; ++i) {                                           /* :314-324 */
        if (isinf(cost[i])) continue;
        if (theta[i] < c->theta_min) c->theta_min = theta[i];
        else if (theta[i] > c->theta_max) c->theta_max = theta[i];
    }
    pair_t *pr = (pair_t *)malloc(sizeof(pair_t) * B);                          /* :326-330 */
    for (int64_t i = 0; i < B; ++i) { pr[i].theta = theta[i]; pr[i].cost = cost[i]; pr[i].idx = i; }
    stable_sort_pairs(pr, B);
    double sum = 0;
    for (int64_t i = 0; i < c->num_elite; ++i) sum += pr[i].theta;
    double mu_new = sum / (double)c->num_elite;
    double ss = 0;
    for (int64_t i = 0; i < c->num_elite; ++i) ss += (pr[i].theta - mu_new) * (pr[i].theta - mu_new);
    double sigma_new = sqrt(ss / (double)c->num_elite);
    c->mu = mu_new; c->sigma = sigma_new;                                       /* :334 */
    free(pr);
}

/* step!  -- :252-335 */
int orc_ce_step(orc_ce *c, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                double *theta_out, double *cost_out) {
    int64_t B = c->num_samples;
    double *theta = (double *)malloc(sizeof(double) * B), *cost = (double *)malloc(sizeof(double) * B);
    int rc = 0;
    c->iter_current++;
    for (int redraw = 0;; ++redraw) {                                           /* :265 */
        if (redraw > 1000) { rc = -2; goto done; }      /* reference would spin (App. B.11) */
        if (c->iter_current == 1) rc = orc_ce_get_positive_samples(c, c->mu_init, c->sigma_init, B, theta);  /* :273 */
        else rc = orc_ce_get_positive_samples(c, c->mu, c->sigma, B, theta);    /* :278 */
        if (rc) goto done;
        orc_compute_value_batch(p, &c->ileqg, x0, u, theta, B, cost, NULL, NULL, NULL, c->nthreads);
        c->n_solves += B; if (redraw) c->n_redraws++;
        for (int64_t i = 0; i < B; ++i) cost[i] = cost[i] + kl_bound / theta[i];   /* :193 */
        int64_t num_inf = 0;
        for (int64_t i = 0; i < B; ++i) num_inf += isinf(cost[i]) ? 1 : 0;       /* :291 */
        int64_t num_valid = B - num_inf;
        double thresh = fmax((double)c->num_elite, (double)B * c->lambda);
        if (c->iter_current == 1 && (double)num_valid < thresh) {                /* :293 */
            c->mu_init *= c->lambda; c->sigma_init *= c->lambda;
        } else if (c->iter_current == 1 && num_valid == B) {                     /* :299 */
            c->mu_init /= c->lambda; c->sigma_init /= c->lambda;
            break;
        } else if ((double)num_valid >= thresh) {                                /* :306 */
            break;
        }
    }
    orc_ce_elite_update(c, theta, cost);                                        /* :314-334 */
    if (theta_out) memcpy(theta_out, theta, sizeof(double) * B);
    if (cost_out) memcpy(cost_out, cost, sizeof(double) * B);
done:
    free(theta); free(cost);
    return rc;
}

/* solve!  -- :364-415 */
int orc_ce_solve(orc_ce *c, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                 double *theta_opt_out, double *x, double *l, double *L, double *value,
                 double *theta_min_out, double *theta_max_out) {
    if (!(kl_bound >= 0)) return -3;                                            /* :368 */
    orc_ce_initialize(c);                                                       /* :369 */
    c->n_final_retries = 0;
    double theta_opt, theta_min = 0.0, theta_max = 0.0;
    if (kl_bound > 0) {
        while (c->iter_current < c->iter_max) {                                 /* :371 */
            int rc = orc_ce_step(c, p, x0, u, kl_bound, NULL, NULL);
            if (rc) return rc;
        }
        theta_min = c->theta_min; theta_max = c->theta_max;                     /* :374 */
        theta_opt = c->use_theta_max ? theta_max : c->mu;                       /* :375-382 */
    } else {
        theta_opt = 0.0;                                                        /* :388 */
    }
    for (int tries = 0;; ++tries) {                                             /* :390 */
        if (tries > 10000) return -4;                   /* reference would spin (App. B.15) */
        orc_solver *s = orc_solver_new(p, &c->ileqg);
        int rc = orc_solve(s, p, x0, u, theta_opt);
        if (rc == ORC_OK || rc == ORC_ITER_MAX) {
            int n = p->n, m = p->m, N = p->N;
            if (x) memcpy(x, s->x, sizeof(double) * n * (N + 1));
            if (l) memcpy(l, s->l, sizeof(double) * m * N);
            if (L) memcpy(L, s->L, sizeof(double) * m * n * N);
            *theta_opt_out = theta_opt;
            if (kl_bound > 0) { *value = s->value_current + kl_bound / theta_opt; *theta_min_out = theta_min; *theta_max_out = theta_max; }  /* :406 */
            else { *value = s->value_current; *theta_min_out = 0.0; *theta_max_out = 0.0; }                                               /* :408 */
            orc_solver_free(s);
            return 0;
        }
        orc_solver_free(s);
        theta_opt = fmax(0.0, theta_opt - c->sigma);                            /* :412 */
        c->n_final_retries++;
    }
}

/* ------------------------------------------------------------------------------------------
 * NelderMeadBilevelOptimizationSolver (RAT iLQR++)  -- nelder_mead_bilevel_optimization.jl
 * ---------------------------------------------------------------------------------------- */
void orc_nm_default(orc_nm *s) {                                                /* :102-128 */
    memset(s, 0, sizeof(*s));
    orc_default_opts(&s->ileqg);
    s->alpha = 1.0; s->beta = 2.0; s->gamma = 0.5; s->eps = 1e-2; s->lambda = 0.5; s->iter_max = 100;
    s->theta_high_init = 3.0; s->theta_low_init = 1e-8;
    s->theta_high = s->theta_high_init; s->theta_low = s->theta_low_init;
    s->has_c_high = 0; s->has_c_low = 0;
}
void orc_nm_initialize(orc_nm *s) {                                             /* :164-168 (c_high/c_low untouched) */
    s->iter_current = 0; s->theta_low = s->theta_low_init; s->theta_high = s->theta_high_init;
}
double orc_nm_compute_cost(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double theta, double kl_bound) {  /* :134-158 */
    orc_solver *sv = orc_solver_new(p, &s->ileqg);
    double cost = INFINITY;
    if (sv) {
        int rc = orc_solve(sv, p, x0, u, theta);
        if (rc == ORC_OK || rc == ORC_ITER_MAX) cost = sv->value_current + kl_bound / theta;
        orc_solver_free(sv);
    }
    s->n_solves++;
    return cost;
}
void orc_nm_step(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double kl_bound) {   /* :174-252 */
    s->iter_current++;
    if (s->c_high < s->c_low) {                                                 /* :184-187 */
        double t = s->theta_low; s->theta_low = s->theta_high; s->theta_high = t;
        t = s->c_low; s->c_low = s->c_high; s->c_high = t;
    }
    const double th_m = s->theta_low;
    double th_r = th_m + s->alpha * (th_m - s->theta_high);                     /* reflection :195 */
    th_r = fmax(s->theta_low_init, th_r);
    const double c_r = orc_nm_compute_cost(s, p, x0, u, th_r, kl_bound);
    if (c_r < s->c_low) {
        double th_e = th_m + s->beta * (th_r - th_m);                           /* expansion :204 */
        th_e = fmax(s->theta_low_init, th_e);
        const double c_e = orc_nm_compute_cost(s, p, x0, u, th_e, kl_bound);
        if (c_e < c_r) { s->theta_high = th_e; s->c_high = c_e; }
        else { s->theta_high = th_r; s->c_high = c_r; }
    } else {
        if (c_r < s->c_high) { s->theta_high = th_r; s->c_high = c_r; }        /* :227-230 */
        double th_c = th_m + s->gamma * (s->theta_high - th_m);                 /* contraction :232 */
        th_c = fmax(s->theta_low_init, th_c);
        const double c_c = orc_nm_compute_cost(s, p, x0, u, th_c, kl_bound);
        if (c_c > s->c_high) {                                                  /* shrink :238-240 */
            s->theta_high = (s->theta_high + s->theta_low) / 2;
            s->c_high = orc_nm_compute_cost(s, p, x0, u, s->theta_high, kl_bound);
        } else { s->theta_high = th_c; s->c_high = c_c; }
    }
}
int orc_nm_solve(orc_nm *s, const orc_problem *p, const double *x0, const double *u, double kl_bound,
                 double *theta_opt, double *x, double *l, double *L, double *value) {      /* :276-352 */
    if (!(kl_bound >= 0)) return -3;
    orc_nm_initialize(s);
    double th_opt;
    if (kl_bound > 0) {
        if (!s->has_c_high) {                                                   /* :283-293 */
            for (int guard = 0;; ++guard) {
                if (guard > 2000) return -4;
                s->c_high = orc_nm_compute_cost(s, p, x0, u, s->theta_high, kl_bound);
                s->has_c_high = 1;
                if (!isinf(s->c_high)) break;
                s->theta_high *= s->lambda; s->theta_high_init *= s->lambda;
            }
        }
        if (!s->has_c_low) {                                                    /* :294-304 */
            for (int guard = 0;; ++guard) {
                if (guard > 2000) return -4;
                s->c_low = orc_nm_compute_cost(s, p, x0, u, s->theta_low, kl_bound);
                s->has_c_low = 1;
                if (!isinf(s->c_low)) break;
                s->theta_low *= s->lambda; s->theta_low_init *= s->lambda;
            }
        }
        for (;;) {                                                              /* :306-324 */
            orc_nm_step(s, p, x0, u, kl_bound);
            const double c_mean = (s->c_low + s->c_high) / 2;
            const double stdev = sqrt(0.5 * ((s->c_high - c_mean) * (s->c_high - c_mean) + (s->c_low - c_mean) * (s->c_low - c_mean)));
            if (stdev < s->eps) break;
            if (s->iter_current == s->iter_max) break;
        }
        th_opt = s->theta_low;                                                  /* :325 */
    } else {
        th_opt = 0.0;
    }
    orc_solver *sv = orc_solver_new(p, &s->ileqg);
    int rc = orc_solve(sv, p, x0, u, th_opt);                                   /* :346 (not in a try) */
    if (rc == ORC_OK || rc == ORC_ITER_MAX) {
        const int n = p->n, m = p->m, N = p->N;
        if (x) memcpy(x, sv->x, sizeof(double) * n * (N + 1));
        if (l) memcpy(l, sv->l, sizeof(double) * m * N);
        if (L) memcpy(L, sv->L, sizeof(double) * m * n * N);
        *theta_opt = th_opt;
        *value = (kl_bound > 0) ? sv->value_current + kl_bound / th_opt : sv->value_current;     /* :347-351 */
    }
    orc_solver_free(sv);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * PETS (pets.jl) on the generative LQ family, serial semantics with injected streams
 * ---------------------------------------------------------------------------------------- */
static double gen_cost(const orc_gen_problem *p, int k, const double *x, const double *u) {
    double c = 0.0;
    model_c(&p->lq, k, x, u, &c);
    if (p->l1u != 0.0) { double a = 0; for (int i = 0; i < p->lq.m; ++i) a += fabs(u[i]); c += p->l1u * a; }
    return c;
}
static void gen_step(const orc_gen_problem *p, const double *x, const double *u, int use_true, const double *zn, double zu, double *xn) {
    const int n = p->lq.n;
    model_f(&p->lq, x, u, xn);
    if (use_true && zu < p->tw2) {                                   /* second mixture component */
        for (int i = 0; i < n; ++i) { double a = p->tmean2[i]; for (int j = 0; j <= i; ++j) a += p->tchol2[IDX(i, j, n)] * zn[j]; xn[i] += a; }
    } else if (p->noise_kind == 0) {
        for (int i = 0; i < n; ++i) { double a = p->nmean[i]; for (int j = 0; j <= i; ++j) a += p->nchol[IDX(i, j, n)] * zn[j]; xn[i] += a; }
    } else {
        for (int i = 0; i < n; ++i) xn[i] += p->nlo + (p->nhi - p->nlo) * zn[i];
    }
}
/* compute_cost_serial  -- pets.jl:128-157 */
int orc_pets_compute_cost(const orc_gen_problem *p, const double *x0, const double *controls, int64_t S, int64_t K,
                          int use_true_model, const double *zn, const double *zu, double *cost) {
    const int n = p->lq.n, m = p->lq.m, N = p->lq.N;
    double x[MAXD], xn[MAXD];
    for (int64_t ii = 0; ii < S; ++ii) {
        double sum = 0.0;
        for (int64_t kk = 0; kk < K; ++kk) {
            const int64_t j = ii * K + kk;
            memcpy(x, x0, sizeof(double) * n);
            double c = 0.0;
            for (int t = 0; t < N; ++t) {
                const double *u = controls + ((size_t)ii * N + t) * m;
                c += gen_cost(p, t, x, u);                                               /* :143 */
                gen_step(p, x, u, use_true_model, zn + ((size_t)j * N + t) * n, zu ? zu[(size_t)j * N + t] : 1.0, xn);   /* :144 */
                memcpy(x, xn, sizeof(double) * n);
            }
            double h; model_h(&p->lq, x, &h);
            c += h;                                                                      /* :147 */
            sum += c;
        }
        cost[ii] = sum / (double)K;                                                      /* mean :150 */
    }
    return 0;
}
void orc_pets_initialize(orc_pets *s) {                                                  /* :70-74 */
    s->iter_current = 0;
    memcpy(s->mu, s->mu_init, sizeof(double) * s->N * s->m);
    memcpy(s->Sigma, s->Sigma_init, sizeof(double) * s->N * s->m * s->m);
}
void orc_pets_update(orc_pets *s, const double *controls, const double *cost, int64_t *elite_idx) {   /* :159-191 */
    const int64_t S = s->num_control_samples, E = s->num_elite, N = s->N, m = s->m;
    pair_t *pr = (pair_t *)malloc(sizeof(pair_t) * S);
    for (int64_t i = 0; i < S; ++i) { pr[i].theta = 0; pr[i].cost = cost[i]; pr[i].idx = i; }
    stable_sort_pairs(pr, S);                                                            /* sort(by = cost) :167 */
    for (int64_t e = 0; e < E; ++e) if (elite_idx) elite_idx[e] = pr[e].idx;
    const double sf = s->smoothing_factor;
    for (int64_t t = 0; t < N; ++t) {
        for (int64_t a = 0; a < m; ++a) {
            double mean = 0;
            for (int64_t e = 0; e < E; ++e) mean += controls[((size_t)pr[e].idx * N + t) * m + a];
            mean /= (double)E;                                                           /* mean :183 */
            double var = 0;
            for (int64_t e = 0; e < E; ++e) { double d = controls[((size_t)pr[e].idx * N + t) * m + a] - mean; var += d * d; }
            var /= (double)(E - 1);                                                      /* var (unbiased) :184 */
            s->mu[t * m + a] = (1.0 - sf) * mean + sf * s->mu[t * m + a];                 /* :186 */
            for (int64_t b = 0; b < m; ++b) {                                            /* Diagonal(var) :184, :187 */
                double *Sg = &s->Sigma[(size_t)t * m * m + a + m * b];
                *Sg = (1.0 - sf) * (a == b ? var : 0.0) + sf * *Sg;
            }
        }
    }
    free(pr);
}
/* lower Cholesky of a small SPD matrix (MvNormal sampling: mu + L z) */
static int chol_lower(int n, const double *A, double *Lo) {
    memset(Lo, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double d = A[IDX(j, j, n)];
        for (int k = 0; k < j; ++k) d -= Lo[IDX(j, k, n)] * Lo[IDX(j, k, n)];
        if (!(d > 0.0)) return 0;
        Lo[IDX(j, j, n)] = sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double v = A[IDX(i, j, n)];
            for (int k = 0; k < j; ++k) v -= Lo[IDX(i, k, n)] * Lo[IDX(j, k, n)];
            Lo[IDX(i, j, n)] = v / Lo[IDX(j, j, n)];
        }
    }
    return 1;
}
int orc_pets_step(orc_pets *s, const orc_gen_problem *p, const double *x0, int use_true_model,
                  const double *zc, const double *zn, const double *zu, double *controls_out, double *cost_out) {   /* :193-245 */
    const int64_t S = s->num_control_samples, N = s->N, m = s->m;
    s->iter_current++;
    double *controls = (double *)malloc(sizeof(double) * S * N * m), *cost = (double *)malloc(sizeof(double) * S);
    double Lc[MAXD * MAXD];
    for (int64_t ii = 0; ii < S; ++ii)                                                   /* :206-216 */
        for (int64_t t = 0; t < N; ++t) {
            if (!chol_lower((int)m, s->Sigma + (size_t)t * m * m, Lc)) { free(controls); free(cost); return -1; }
            const double *z = zc + ((size_t)ii * N + t) * m;
            for (int64_t a = 0; a < m; ++a) {
                double v = s->mu[t * m + a];
                for (int64_t b = 0; b <= a; ++b) v += Lc[IDX(a, b, m)] * z[b];
                controls[((size_t)ii * N + t) * m + a] = v;                               /* rand(rng, MvNormal(mu, Sigma)) */
            }
        }
    orc_pets_compute_cost(p, x0, controls, S, s->num_trajectory_samples, use_true_model, zn, zu, cost);
    orc_pets_update(s, controls, cost, NULL);
    if (controls_out) memcpy(controls_out, controls, sizeof(double) * S * N * m);
    if (cost_out) memcpy(cost_out, cost, sizeof(double) * S);
    free(controls); free(cost);
    return 0;
}

/* x^y of the reference (fdlibm_pow.h) over arrays: tests/test_cpu_pow.py, tests/test_gpu_pow.py */
void orc_pow_array(const double *x, const double *y, long n, double *out) {
    for (long i = 0; i < n; ++i) out[i] = orc_pow(x[i], y[i]);
}
