// wide.h -- argument block of the general-size solve kernel (wide.hip): problems beyond the n <= 12, m <= 4 tile of the MFMA kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

#define WIDE_MAX_N 32
#define WIDE_MAX_M 32

// Device tables of an LQ-family problem at its own size: the caller's column-major arrays as they are (C ABI layout), plus what the
// host derives once per problem (symmetrised Hessians, inv(W(k)), logdet W(k)).
struct WideProblemDev {
    int n, m, N, cost_tv, W_tv;
    double kappa, q0f;
    const double *A, *B;             // [n*n], [n*m]
    const double *Q, *R, *P;         // [Nc][n*n] Symmetric(c_xx), [Nc][m*m] Symmetric(c_uu), [Nc][m*n] c_ux
    const double *qv, *rv, *q0;      // [Nc][n], [Nc][m], [Nc]
    const double *Qf, *qvf;          // [n*n] Symmetric, [n]
    const double *W, *Winv, *ldW;    // [Nw][n*n], [Nw][n*n] Symmetric(inv(W)), [Nw] logdet W(k)
};

struct WideArgs {
    WideProblemDev pb;
    OptsDev op;
    int B;
    const double *x0, *u0, *theta;   // [n], [N*m] column-major (time slowest), [B]
    // per-sample scratch (the solver object of one theta-sample: x_array, l_array and their candidates, L_array, dl)
    double *xs, *us;                 // [B][2][(N+1)*n], [B][2][N*m]
    double *L, *dl;                  // [B][N*m*n] (m x n column-major per step), [B][N*m]
    int *nom;                        // [B] which of the two (x, u) slots holds x_array / l_array when the solve ends
    // outputs (any may be null)
    double *out_value; int *out_status, *out_iters, *out_ls;
    double *out_cost; double kl_bound;
    double *hist; int hist_cap; int *hist_n;     // sample 0's line-search history (eps, value difference) or null
};

size_t wide_lds_bytes(int n, int m);
hipError_t launch_wide_solve(const WideArgs &a, hipStream_t s);
