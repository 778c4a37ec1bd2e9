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
    // The same tables as REGISTER IMAGES for the block form of wide32.h: a block of RT x CT tiles is (RT CT 4) runs of 64 doubles -- tile (a, b),
    // register r, lane (g, j) = element (16 a + 4 r + g, 16 b + j), zero outside the matrix (unit diagonal where noted) -- so a load is one
    // coalesced instruction with an immediate offset: no address clamps, no 0 / 1 masks (which the compiler would otherwise hoist out of the
    // time loop and keep, four per tile, in registers).  NT = tiles of the states, MT = of the controls, as launch_wide_solve picks them.
    int img_nt, img_mt;
    const double *tA, *tAT, *tB, *tBT;     // [NT NT], A' [NT NT], [NT MT], B' natural rows [MT NT]
    const double *tQ, *tP, *tPT, *tR;      // [Nc][NT NT], P natural rows [Nc][MT NT], P' [Nc][NT MT], [Nc][MT MT] unit diagonal beyond m
    const double *tqv, *trv;               // [Nc][NT] / [Nc][MT] one-column blocks (column 0)
    const double *tQf, *tqvf;              // [NT NT], [NT]
    const double *tWinv, *tW;              // [Nw][NT NT] unit diagonal beyond n; [Nw][NT NT]
};
#define WIDE_IMG_TILE 256            /* doubles per tile image */

struct WideArgs {
    WideProblemDev pb;
    OptsDev op;
    int B;
    int fast16;                      // n <= 16, m <= 4: sweeps in registers on the matrix pipe (wide16.h); 0 = the general sweep (A/B, tests)
    int fast32;                      // every other size: sweeps and rollouts in registers in block form (wide32.h); 0 = the general LDS sweep
    const double *x0, *u0, *theta;   // [n], [N*m] column-major (time slowest), [B]
    // per-sample scratch (the solver object of one theta-sample: x_array, l_array and their candidates, L_array, dl)
    double *xs, *us;                 // [B][2][(N+1)*n], [B][2][N*m]
    double *L, *dl;                  // [B][N*m*n] (m x n column-major per step), [B][N*m]
    double *gq, *gr, *gc;            // block form (wide32.h), time-invariant cost: [B][2][N*n] c_x, [B][2][N*m] c_u of the slots' trajectories and
                                     // [B][2][64] the lanes' parts of their summed stage costs (grad32, once per rollout instead of per sweep step)
    int *nom;                        // [B] which of the two (x, u) slots holds x_array / l_array when the solve ends
    // outputs (any may be null)
    double *out_value; int *out_status, *out_iters, *out_ls;
    double *out_cost; double kl_bound;
    double *hist; int hist_cap; int *hist_n;     // sample 0's line-search history (eps, value difference) or null
};

// Operator forms at general size (the reference's building blocks as individual calls; unit parity with test/ileqg_test.jl).  All arrays
// are device pointers in the C ABI layout (column-major, time slowest); `count` workgroups each handle one rollout / one sample.
enum WideOp { WOP_ROLL_OPEN = 1, WOP_ROLL_FEEDBACK, WOP_COST, WOP_NOISY, WOP_APPROX, WOP_DP_GAIN, WOP_DP_EVAL };
struct WideOpArgs {
    WideProblemDev pb;
    OptsDev op;
    int opcode;
    long count;                           // rollouts (WOP_NOISY) or samples (WOP_DP_*), else 1
    // trajectories
    const double *x0, *u;                 // open loop: x0 [n], u [N*m];  WOP_COST / WOP_APPROX: x = xbar [(N+1)*n], u [N*m]
    const double *xbar, *l, *L;           // feedback / noisy: nominal states, l [N*m], L [N*m*n] (noisy: null = open loop)
    double *x_out, *u_out, *cost_out;     // rollouts / noisy: per workgroup [(N+1)*n], [N*m], [1]; any may be null
    const double *z; unsigned long long seed; const double *Wchol;   // noisy: injected N(0,1) [count][N][n] or null (Philox); chol_lower(W(k)) [Nw][n*n]
    // ApproximationResult: outputs of WOP_APPROX, inputs of WOP_DP_* (per sample, sample slowest)
    double *q, *qv, *Q, *r, *R, *P, *A, *B, *W;
    // sweeps
    const double *theta, *mu_in, *dlin;   // [count]; dlin: [N*m] per sample or null (WOP_DP_EVAL)
    double *mu, *delta;                   // [count] in/out (WOP_DP_GAIN)
    double *Lio, *dl_out;                 // L: [N*m*n] per sample, in (eval) / out (gain); dl out (gain)
    double *value; int *status;           // [count]
    double *ds, *dsv, *dS, *dg, *dG, *dH; // DynamicProgrammingResult dumps of sample 0, any may be null
};

size_t wide_lds_bytes(int n, int m);
hipError_t launch_wide_op(const WideOpArgs &a, hipStream_t s);
hipError_t launch_wide_solve(const WideArgs &a, hipStream_t s);
bool wide32_applies(const WideArgs &a);
