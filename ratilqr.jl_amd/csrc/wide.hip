// wide.hip -- the complete solve! of one theta-sample for problems beyond the 12 + 4 tile of kernels.hip (n <= 32, m <= 32, LQ family).
//
// The MFMA kernels of kernels.hip are specialised for n <= 12, m <= 4 (every BASELINE configuration): there [A|B] is one 12 x 16 operand
// and the value function one accumulator tile.  The reference itself takes its dimensions from the arrays (ileqg.jl:229); this kernel
// covers what lies beyond the tile with ONE WORKGROUP (one wavefront) PER SAMPLE and every matrix of a backward step in LDS at its own
// size (column-major with odd leading dimensions, 6 n^2 + 5 n m + 2 m^2 doubles + padding: 112 KB at n = m = 32 of the CU's 160 KB; four
// workgroups per CU up to n = 24, m = 8).  The whole solve! runs inside the
// launch -- initialize! (ileqg.jl:214-236), then step! / line_search! until the convergence or iter_max test (:598-613, :494-592,
// :635-659) -- with the per-sample control flow wave-uniform in registers.  Nothing is linearised into HBM: an LQ-family step's
// derivatives are its tables plus a diagonal (f_x = A + 3 kappa diag(x^2)), so the sweep forms them from (x_t, u_t) as it goes.
//
// One backward step (ileqg.jl:361-391 / :435-460), with M = inv(W) - theta S = U'U (Cholesky; isposdef(M) <=> every pivot > 0):
//     [Z | z] = U^-T [S | s_vec]             one forward substitution, a lane per column
//     D S = S + theta Z'Z,  D s_vec = s_vec + theta Z'z,  s_vec' M^-1 s_vec = z'z,  logdet(W M) = logdet W + 2 sum log U_kk
//     g = r + B' D s_vec,  G = P + B'(D S) A,  H = R + B'(D S) B + mu I;  gain sweep: H = Uh'Uh (isposdef(H), else mu, Delta are
//     raised and the sweep restarts, :372-378), [L | dl] = -H^-1 [G | g]
//     s, s_vec, S as in :383-391.
// Round 4: the matrix products run on the matrix pipe (prod: 16 x 16 blocks, operands read from LDS straight into the MFMA's lanes), both
// factorisations and their substitutions entirely in registers (chol_reg: a lane per column, rows broadcast by v_readlane, no LDS access
// and no barrier per row), the step's tables are fetched once per sweep, and every LDS pointer carries its address space (ds_read /
// ds_write instead of flat accesses).  A wave issues ~5 k instructions per backward step at 16 x 4 (a third of them scalar: loop control
// and edge masks) for ~350 wave-instructions' worth of multiply-adds, and is parked half of its lifetime on the dependent chains of the
// factorisation and on LDS round trips between phases; see DESIGN.md and profiles/r04_wide_sizes.md.
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "wide.h"
#include "device_utils.h"
#include "rat_normal.h"

#ifndef WIDE_PART
#define WIDE_PART 3
#endif

namespace {

// Every matrix of a step lives in LDS, and the pointers say so: through plain `double *` members the compiler can only emit FLAT
// loads and stores (64-bit address arithmetic, the flat path's latency, vmcnt and lgkmcnt both held at zero around each) -- as rounds 1-3
// did; with the address space in the type they are ds_read_b64 / ds_write_b64 off a 32-bit offset.
typedef __attribute__((address_space(3))) double ldsd;
struct Ws {
    ldsd *S, *U, *Z, *DS, *T, *At, *Bm, *F, *G, *Lt, *H;
    ldsd *sv, *z, *dsv, *qv, *sv0, *xt, *xb, *g, *dlv, *rv, *ut, *hv;
    ldsd *Qs, *Ps, *Rs, *qvs, *rvs, *dA;     // the step's tables staged in LDS (stage_cost_tables): Q, P, R, q_vec, r_vec; diag(A)
};

// sum over the wavefront, the same bits on every lane: DPP rotations inside the 16-lane rows, then the four row sums through scalar
// registers (a butterfly of __shfl_xor goes through the LDS crossbar six times: ~5x the latency, and this sits on every row of the
// lockstep factorisation below)
__device__ __forceinline__ double wsum(double v) {
    const double r = row_sum16(v);
    return (readlane_f64(r, 0) + readlane_f64(r, 16)) + (readlane_f64(r, 32) + readlane_f64(r, 48));
}

// ---- small dense linear algebra on matrices in LDS, one wavefront ------------------------------------------------------------------
// Lanes are an 8 x 8 grid (li, lj) over matrix entries: no integer division anywhere, and products are register-blocked 2 x 2
// (entries (i, j), (i + 8, j), (i, j + 8), (i + 8, j + 8): four LDS reads per four FMAs).
template <class F> __device__ __forceinline__ void each(const int rows, const int cols, F f) {
    const int li = threadIdx.x & 7, lj = threadIdx.x >> 3;
    for (int j = lj; j < cols; j += 8)
        for (int i = li; i < rows; i += 8) f(i, j);
}
// C(i, j) = sum_k A(i, k) B(k, j), A(i, k) = pa(i)[k * sa], B(k, j) = pb(j)[k * sb]; store(i, j, c) for every entry.
// On the matrix pipe: C is cut into 16 x 16 blocks, a block is ceil(K / 4) v_mfma_f64_16x16x4 whose operands come straight from LDS --
// lane (g, j) of slice s loads A(16 rb + j, 4 s + g) and B(4 s + g, 16 cb + j), zero outside the matrices (nothing in LDS is padded to
// the block size), one ds_read_b64 each -- and leaves rows 4 r + g of column j in accumulator register r.  1024 multiply-adds per
// instruction and two LDS reads, against one LDS read per multiply-add in the round-3 vector loop (which ran at ~130 cycles per
// multiply-add per lane at one wavefront per SIMD: 16 x 16 x 16 took 8.6 k cycles; this takes ~0.6 k).  The A slices of a row block
// are loaded once for all its column blocks.
template <int NS, class PA, class PB, class ST>
__device__ __forceinline__ void prod_ns(const int rows, const int cols, const int K, PA pa, const int sa, PB pb, const int sb, ST store) {
    const int lane = threadIdx.x, j16 = lane & 15, g = lane >> 4;
    for (int r0 = 0; r0 < rows; r0 += 16) {
        const int ia = r0 + j16;
        const bool va = ia < rows;
        const auto ap = pa(va ? ia : 0);
        double av[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {                      // (the address is clamped into the matrix, the value masked after the load:
            const int kk = 4 * sl + g;                         //  no branch around any load)
            const double v = ap[((kk < K) ? kk : 0) * sa];
            av[sl] = (va && kk < K) ? v : 0.0;
        }
        for (int c0 = 0; c0 < cols; c0 += 16) {
            const int jb = c0 + j16;
            const bool vb = jb < cols;
            const auto bp = pb(vb ? jb : 0);
            double bv[NS];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int kk = 4 * sl + g;
                const double v = bp[((kk < K) ? kk : 0) * sb];
                bv[sl] = (vb && kk < K) ? v : 0.0;
            }
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) acc = MFMA(av[sl], bv[sl], acc);
            if (vb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int i = r0 + 4 * r + g; if (i < rows) store(i, jb, acc[r]); }
            }
        }
    }
}
template <class PA, class PB, class ST>
__device__ __forceinline__ void prod(const int rows, const int cols, const int K, PA pa, const int sa, PB pb, const int sb, ST store) {
    if (K <= 4) { prod_ns<1>(rows, cols, K, pa, sa, pb, sb, store); return; }
    // K > 4: one body for every depth (slices beyond K are skipped by uniform branches) -- a body per depth tripled the code of each of
    // the nine call sites, and a backward step already runs past the instruction cache
    const int lane = threadIdx.x, j16 = lane & 15, g = lane >> 4;
    const int ns = (K + 3) >> 2;                               // <= 8 slices
    for (int r0 = 0; r0 < rows; r0 += 16) {
        const int ia = r0 + j16;
        const bool va = ia < rows;
        const auto ap = pa(va ? ia : 0) + g * sa;
        double av[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) av[sl] = (va && sl < ns && 4 * sl + g < K) ? ap[(4 * sl) * sa] : 0.0;
        for (int c0 = 0; c0 < cols; c0 += 16) {
            const int jb = c0 + j16;
            const bool vb = jb < cols;
            const auto bp = pb(vb ? jb : 0) + g * sb;
            double bv[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) bv[sl] = (vb && sl < ns && 4 * sl + g < K) ? bp[(4 * sl) * sb] : 0.0;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) if (sl < ns) acc = MFMA(av[sl], bv[sl], acc);
            if (vb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int i = r0 + 4 * r + g; if (i < rows) store(i, jb, acc[r]); }
            }
        }
    }
}

// Cholesky factorisation X = U'U of the k x k matrix Um (k <= K <= 32; its upper triangle is read) FUSED with the forward substitutions
// U'Y = R for nrhs <= 32 right-hand-side columns (read from Rs, solution to Rm), one more right-hand side `vec`, and -- when `back` -- the back
// substitutions U X = Y of all of them, ENTIRELY IN REGISTERS: lane c < k owns column c of the matrix, lane 32 + c column c of R, each as
// K doubles.  Right-looking over the rows: row i is scaled by 1 / U_ii, then every later row q of every column loses U_iq U_ic -- U_iq is
// lane q's entry of the scaled row, broadcast by v_readlane, so a row costs K - i - 1 (readlane pair + FMA) and no LDS access, no
// barrier; each entry still is M_ij - sum_k U_ki U_kj in ascending k like the textbook loop.  (Round 3 kept the columns in LDS and ran
// the left-looking loop with a barrier per row: ~1,400 cycles per row at either size; this form is ~250.)  false <=> a pivot <= 0 or
// NaN: what LAPACK potrf reports and isposdef tests.  yv: lane i < k ends with entry i of the extra right-hand side's solution;
// sumlog = sum_i log U_ii.  The factor itself is not written back: nothing reads it.
// r = sqrt(d), ri = 1 / r for a pivot d > 0: v_rsq_f64 (~2^-26) and two Newton steps on the reciprocal root, r = d ri with one
// correction -- both to an ulp or two, which is all a Cholesky pivot needs (nothing here is compared bit for bit) -- instead of the
// IEEE sqrt expansion followed by a reciprocal: ~60 cycles of the row's serial chain instead of ~280.
__device__ __forceinline__ void pivot_root(const double d, double &r, double &ri) {
    if (!(d < 1e300) || d < 1e-290) { r = sqrt(d); ri = 1.0 / r; return; }      // (outside the range the plain iteration is safe in)
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    y = y * (1.5 - h * (y * y));
    y = y * (1.5 - h * (y * y));
    double s0 = d * y;
    s0 = fma(fma(-s0, s0, d), 0.5 * y, s0);
    r = s0; ri = y;
}
template <int K>
__device__ __noinline__ bool chol_reg(const ldsd *Um, const int k, const int ld, const ldsd *Rs, ldsd *Rm, const int nrhs, const int ldr, const ldsd *vec,
                                      const bool back, double &yv, double &sumlog) {
    const int lane = threadIdx.x;
    const bool isM = lane < k, isR = lane >= 32 && lane - 32 < nrhs;
    const ldsd *const srcp = isM ? Um + ld * lane : Rs + ldr * (isR ? lane - 32 : 0);        // (every column is read before any is written:
    ldsd *const colp = Rm + ldr * (isR ? lane - 32 : 0);                                         //  the solution may overwrite the matrix)
    double col[K];
#pragma unroll
    for (int q = 0; q < K; ++q) col[q] = ((isM ? q <= lane : isR) && q < k) ? srcp[q] : 0.0;      // (upper triangle of the matrix)
    double accy = isM ? vec[lane] : 0.0;                  // lane c: vec_c - sum_{q < i} U_qc y_q so far; y_c once row c is done
    double myr = 1.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (i < k) {
            const double d = readlane_f64(col[i], i);
            if (!(d > 0.0)) return false;
            double r, ri;
            pivot_root(d, r, ri);
            const double ui = col[i] * ri;                // U_ic on the matrix lanes, Y_ic on the right-hand-side lanes
            col[i] = (lane == i) ? r : ui;
            if (lane == i) myr = r;
            const double yi = readlane_f64(accy, i) * ri;
            accy = (lane == i) ? yi : ((lane > i) ? fma(-ui, yi, accy) : accy);
#pragma unroll
            for (int q = i + 1; q < K; ++q) {
                const double uq = readlane_f64(ui, q);    // U_iq (0 beyond the matrix: those lanes hold zero columns)
                col[q] = fma(-uq, ui, col[q]);
            }
        }
    }
    sumlog = wsum(isM ? log(myr) : 0.0);
    if (back) {
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            if (i < k) {
                const double ri = fast_rcp(readlane_f64(col[i], i));
                const double dot = wsum((isM && lane > i) ? col[i] * accy : 0.0);          // sum_{q > i} U_iq x_q of the extra right-hand side
                if (lane == i) accy = (accy - dot) * ri;
                const double xi = isR ? col[i] * ri : 0.0;                                 // (matrix lanes keep the factor: their update is - U_qi * 0)
                if (isR) col[i] = xi;
#pragma unroll
                for (int q = 0; q < i; ++q) col[q] = fma(-readlane_f64(col[q], i), xi, col[q]);
            }
        }
    }
    yv = accy;
    if (isR) {
#pragma unroll
        for (int q = 0; q < K; ++q) if (q < k) colp[q] = col[q];
    }
    __syncthreads();
    return true;
}
__device__ __forceinline__ bool chol_solve(const ldsd *Um, const int k, const int ld, const ldsd *Rs, ldsd *Rm, const int nrhs, const int ldr, const ldsd *vec,
                                           const bool back, double &yv, double &sumlog) {
    if (k <= 4) return chol_reg<4>(Um, k, ld, Rs, Rm, nrhs, ldr, vec, back, yv, sumlog);
    if (k <= 8) return chol_reg<8>(Um, k, ld, Rs, Rm, nrhs, ldr, vec, back, yv, sumlog);
    if (k <= 16) return chol_reg<16>(Um, k, ld, Rs, Rm, nrhs, ldr, vec, back, yv, sumlog);
    if (k <= 24) return chol_reg<24>(Um, k, ld, Rs, Rm, nrhs, ldr, vec, back, yv, sumlog);
    return chol_reg<32>(Um, k, ld, Rs, Rm, nrhs, ldr, vec, back, yv, sumlog);
}

// Where a step's quadratic model comes from: the solver forms it on the fly from the trajectory (x, u) and the problem tables; the
// operator forms (rat_dp_*) read caller-built ApproximationResult arrays in the C ABI layout (A != nullptr).
struct Tiles {
    const double *x, *u;
    const double *q, *qv, *Q, *r, *R, *P, *A, *B;
};
struct Dump { double *s, *sv, *S, *g, *G, *H; };        // DynamicProgrammingResult arrays of the operator forms (any may be null)

// The cost tables a backward step reads -- Q, P, R, q_vec, r_vec -- from global memory into LDS in one burst of independent
// loads: once per sweep when they do not depend on the step, at the top of the step otherwise (time-varying cost / covariance, caller-
// built tiles).  Round 3 read them where they were used -- inside the store of a product, inside a dot product -- each time behind its
// own ~2 k cycles of exposed latency at one wavefront per SIMD: five to seven such waits per step.
__device__ __forceinline__ void stage_cost_tables(const WideProblemDev &pb, Ws &w, const double *Qk, const double *Rk, const double *Pk,
                                                  const double *qvk, const double *rvk) {
    const int n = pb.n, m = pb.m, lane = threadIdx.x, ldn = n | 1, ldm = m | 1;
    each(n, n, [&](int i, int j) { w.Qs[i + ldn * j] = Qk[i + n * j]; });
    each(m, n, [&](int g, int j) { w.Ps[g + ldm * j] = Pk[g + m * j]; });
    each(m, m, [&](int g, int g2) { w.Rs[g + ldm * g2] = Rk[g + m * g2]; });
    if (qvk && lane < n) w.qvs[lane] = qvk[lane];
    if (rvk && lane >= 32 && lane - 32 < m) w.rvs[lane - 32] = rvk[lane - 32];
}
// the entries of an n x n matrix (n <= 32) this lane visits in each(n, n, .), with a compile-time running index: f(i, j, e), e < 16
template <class F> __device__ __forceinline__ void each16(const int n, F f) {
    const int li = threadIdx.x & 7, lj = threadIdx.x >> 3;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { const int i = li + 8 * ii, j = lj + 8 * jj; if (i < n && j < n) f(i, j, 4 * jj + ii); }
}

// solve_approximate_dp (gain = false, :412-465) / one pass of solve_approximate_dp! (gain = true, :341-406) over the trajectory.
// Returns 0, 2 (M not positive definite) or -1 (H not positive definite: the caller raises mu and restarts).
// (The problem block, the LDS map and the tile pointers are taken into locals: through references of a function that is really called --
//  this one is, from five places -- every pb.n, w.S, tl.x is a load through a generic pointer with a full wait behind it, hundreds per step.)
__device__ int sweep(const WideProblemDev &pb_in, const Ws &w_in, const Tiles &tl_in, const double theta, const double mu,
                     const bool gain, const bool zeroL, double *Lg, double *dlg, const double *dlin, const Dump *dump_in, double &value) {
    const WideProblemDev pb = pb_in;
    Ws w = w_in;
    const Tiles tl = tl_in;
    Dump dump_v = {};
    if (dump_in) dump_v = *dump_in;
    const Dump *const dump = dump_in ? &dump_v : nullptr;
    const int n = pb.n, m = pb.m, N = pb.N, n2 = n * n, nm = n * m, mm = m * m, lane = threadIdx.x;
    const int ldn = n | 1, ldm = m | 1;          // odd leading dimensions in LDS: a column stride of n doubles puts every lane on one bank
    const bool arr = tl.A != nullptr;
    const double *const x = tl.x, *const u = tl.u;
    // terminal condition (:352-354 / :429-431)
    double s1;
    if (arr) {
        const double *Qn = tl.Q + (size_t)N * n2;
        each(n, n, [&](int i, int j) { w.S[i + ldn * j] = (i <= j) ? Qn[i + n * j] : Qn[j + n * i]; });
        if (lane < n) w.sv[lane] = tl.qv[(size_t)N * n + lane];
        s1 = tl.q[N];
    } else {
        if (lane < n) w.xt[lane] = x[(size_t)N * n + lane];
        __syncthreads();
        each(n, n, [&](int i, int j) { w.S[i + ldn * j] = pb.Qf[i + n * j]; });
        double part0 = 0.0;
        if (lane < n) {
            double acc = 0.0;
            #pragma unroll 8
            for (int j = 0; j < n; ++j) acc += pb.Qf[lane + n * j] * w.xt[j];
            w.sv[lane] = acc + pb.qvf[lane];
            part0 = w.xt[lane] * (0.5 * acc + pb.qvf[lane]);
        }
        s1 = wsum(part0) + pb.q0f;
    }
    double part = 0.0;
    __syncthreads();
    if (dump) {
        if (dump->s && lane == 0) dump->s[N] = s1;
        if (dump->sv && lane < n) dump->sv[(size_t)N * n + lane] = w.sv[lane];
        if (dump->S) each(n, n, [&](int i, int j) { dump->S[(size_t)N * n2 + i + n * j] = w.S[i + ldn * j]; });
    }
    // what does not depend on the step is fetched once: the cost tables (LDS), inv(W) (registers), and -- LQ family -- A itself, whose diagonal alone
    // changes from step to step (f_x = A + 3 kappa diag(x^2))
    const bool cost_per_step = arr || pb.cost_tv, w_per_step = pb.W_tv != 0;
    if (!cost_per_step) stage_cost_tables(pb, w, pb.Q, pb.R, pb.P, pb.qv, pb.rv);
    double wreg[16];                                       // inv(W)'s entries of this lane (each16): no LDS copy, no load per step
    if (!w_per_step) each16(n, [&](int i, int j, int e) { wreg[e] = pb.Winv[i + n * j]; });
    if (!arr) {
        each(n, n, [&](int i, int j) { w.At[i + ldn * j] = pb.A[i + n * j]; });
        if (lane < n) w.dA[lane] = pb.A[lane + n * lane];
    }
    // (x_t, u_t) of the step after this one are fetched while this one runs
    double xpre = 0.0, upre = 0.0;
    if (!arr) {
        if (lane < n) xpre = x[(size_t)(N - 1) * n + lane];
        if (lane >= 32 && lane - 32 < m) upre = u[(size_t)(N - 1) * m + lane - 32];
    }
    __syncthreads();
    for (int t = N - 1; t >= 0; --t) {
        const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
        const double *Wk = pb.W + (size_t)kw * n2;
        if (cost_per_step) {
            if (arr) stage_cost_tables(pb, w, tl.Q + (size_t)t * n2, tl.R + (size_t)t * mm, tl.P + (size_t)t * nm, nullptr, nullptr);
            else stage_cost_tables(pb, w, pb.Q + (size_t)kc * n2, pb.R + (size_t)kc * mm, pb.P + (size_t)kc * nm, pb.qv + (size_t)kc * n, pb.rv + (size_t)kc * m);
        }
        if (w_per_step) { const double *Wik = pb.Winv + (size_t)kw * n2; each16(n, [&](int i, int j, int e) { wreg[e] = Wik[i + n * j]; }); }
        if (!gain) {
            const double *Lt_g = Lg + (size_t)t * nm;
            each(m, n, [&](int g, int j) { w.Lt[g + ldm * j] = zeroL ? 0.0 : Lt_g[g + m * j]; });
            if (lane < m) w.dlv[lane] = dlin ? dlin[(size_t)t * m + lane] : 0.0;
        }
        double q;
        if (arr) {                                                                         // caller-built tiles of step t
            const double *At_g = tl.A + (size_t)t * n2, *Bt_g = tl.B + (size_t)t * nm;
            each(n, n, [&](int i, int j) { w.At[i + ldn * j] = At_g[i + n * j]; });
            each(n, m, [&](int i, int g) { w.Bm[i + ldn * g] = Bt_g[i + n * g]; });
            if (lane < n) w.qv[lane] = tl.qv[(size_t)t * n + lane];
            if (lane < m) w.rv[lane] = tl.r[(size_t)t * m + lane];
            q = tl.q[t];
            __syncthreads();
        } else {                                                                           // approximate_model at (x_t, u_t)  (:294-313)
            if (lane < n) { w.xt[lane] = xpre; w.At[lane + ldn * lane] = w.dA[lane] + 3.0 * pb.kappa * (xpre * xpre); }
            if (lane >= 32 && lane - 32 < m) w.ut[lane - 32] = upre;
            if (t > 0) {
                if (lane < n) xpre = x[(size_t)(t - 1) * n + lane];
                if (lane >= 32 && lane - 32 < m) upre = u[(size_t)(t - 1) * m + lane - 32];
            }
            __syncthreads();
            part = 0.0;
            if (lane < n) {
                double qx = 0.0, pu = 0.0;
                #pragma unroll 8
                for (int j = 0; j < n; ++j) qx += w.Qs[lane + ldn * j] * w.xt[j];
                #pragma unroll 8
                for (int g = 0; g < m; ++g) pu += w.Ps[g + ldm * lane] * w.ut[g];
                const double qvl = w.qvs[lane];
                w.qv[lane] = qx + pu + qvl;
                part = w.xt[lane] * (0.5 * qx + qvl);
            } else if (lane >= 32 && lane - 32 < m) {
                const int g = lane - 32;
                double ru = 0.0, px = 0.0;
                #pragma unroll 8
                for (int g2 = 0; g2 < m; ++g2) ru += w.Rs[g + ldm * g2] * w.ut[g2];
                #pragma unroll 8
                for (int j = 0; j < n; ++j) px += w.Ps[g + ldm * j] * w.xt[j];
                const double rvl = w.rvs[g];
                w.rv[g] = ru + px + rvl;
                part = w.ut[g] * (0.5 * ru + px + rvl);
            }
            q = wsum(part) + pb.q0[kc];
        }
        // M = inv(W) - theta S  (:365) into U; the forward substitution reads its right-hand sides from S and leaves Z IN U (M is in
        // registers by then)
        each16(n, [&](int i, int j, int e) { w.U[i + ldn * j] = wreg[e] - theta * w.S[i + ldn * j]; });
        __syncthreads();
        double sumlog = 0.0, zreg = 0.0;
        if (!chol_solve(w.U, n, ldn, w.S, w.Z, n, ldn, w.sv, false, zreg, sumlog)) return 2;      // @assert isposdef(M)  :366 / :440;  [Z | z] = U^-T [S | s_vec]
        if (theta == 0.0) {                          // D = I exactly, whatever the size of S: the products below must not see Z'Z
            each(n, n, [&](int i, int j) { w.Z[i + ldn * j] = 0.0; });
            zreg = 0.0;
        }
        if (lane < n) w.z[lane] = zreg;
        __syncthreads();
        prod(n, n, n, [&](int i) { return w.Z + ldn * i; }, 1, [&](int j) { return w.Z + ldn * j; }, 1,
             [&](int i, int j, double c) { w.DS[i + ldn * j] = w.S[i + ldn * j] + theta * c; });          // D S  (:367; S symmetric)
        part = 0.0;
        if (lane < n) {
            double acc = 0.0;
            #pragma unroll 8
            for (int k = 0; k < n; ++k) acc += w.Z[k + ldn * lane] * w.z[k];
            w.dsv[lane] = w.sv[lane] + theta * acc;                                        // D s_vec
            part = zreg * zreg;
        }
        const double zz = wsum(part);                                                      // s_vec' M^-1 s_vec
        __syncthreads();
        prod(n, n + m, n, [&](int i) { return w.DS + i; }, ldn,
             [&](int j) { return (j < n) ? w.At + ldn * j : w.Bm + ldn * (j - n); }, 1,
             [&](int i, int j, double c) { if (j < n) w.T[i + ldn * j] = c; else w.F[i + ldn * (j - n)] = c; });   // T = (D S) A,  F = (D S) B
        __syncthreads();
        prod(m, n, n, [&](int g) { return w.Bm + ldn * g; }, 1, [&](int j) { return w.T + ldn * j; }, 1,
             [&](int g, int j, double c) { w.G[g + ldm * j] = w.Ps[g + ldm * j] + c; });                              // G = P + B'(D S) A  (:369)
        prod(m, m, n, [&](int g) { return w.Bm + ldn * g; }, 1, [&](int g2) { return w.F + ldn * g2; }, 1,
             [&](int g, int g2, double c) {                                                                       // H = Symmetric(R + B'(D S) B + mu I)  (:370-371)
                 if (g <= g2) {
                     const double v = w.Rs[g + ldm * g2] + c + ((g == g2) ? mu : 0.0);
                     w.H[g + ldm * g2] = v; w.H[g2 + ldm * g] = v;
                 }
             });
        if (lane < m) {                                                                    // g = r + B' D s_vec  (:368)
            double acc = 0.0;
            #pragma unroll 8
            for (int k = 0; k < n; ++k) acc += w.Bm[k + ldn * lane] * w.dsv[k];
            w.g[lane] = w.rv[lane] + acc;
        }
        __syncthreads();
        if (dump) {
            if (dump->g && lane < m) dump->g[(size_t)t * m + lane] = w.g[lane];
            if (dump->G) each(m, n, [&](int g, int j) { dump->G[(size_t)t * nm + g + m * j] = w.G[g + ldm * j]; });
            if (dump->H) each(m, m, [&](int g, int g2) { dump->H[(size_t)t * mm + g + m * g2] = w.H[g + ldm * g2]; });
        }
        if (gain) {                                                                        // [L | dl] = -H \ [G | g]  (:379-381)
            each(m, n, [&](int g, int j) { w.Lt[g + ldm * j] = -w.G[g + ldm * j]; });
            if (lane < m) w.hv[lane] = -w.g[lane];
            __syncthreads();
            double dreg = 0.0, dummy = 0.0;
            if (!chol_solve(w.H, m, ldm, w.Lt, w.Lt, n, ldm, w.hv, true, dreg, dummy)) return -1; // !isposdef(H)  :372;  both substitutions
            if (lane < m) { w.dlv[lane] = dreg; dlg[(size_t)t * m + lane] = dreg; }
            double *Lt_g = Lg + (size_t)t * nm;
            each(m, n, [&](int g, int j) { Lt_g[g + m * j] = w.Lt[g + ldm * j]; });
            __syncthreads();
        }
        part = 0.0;
        if (lane < m) {
            double hd = 0.0;
            #pragma unroll 8
            for (int g2 = 0; g2 < m; ++g2) hd += w.H[lane + ldm * g2] * w.dlv[g2];
            w.hv[lane] = hd + w.g[lane];
            part = w.dlv[lane] * (0.5 * hd + w.g[lane]);                                   // 0.5 dl'H dl + dl'g  (:383)
        }
        double s0 = q + s1 + wsum(part);
        if (theta == 0.0) {                                                                // :384-385
            part = 0.0;
            each(n, n, [&](int i, int j) { part += Wk[i + n * j] * w.S[j + ldn * i]; });
            s0 += 0.5 * wsum(part);
        } else {                                                                           // :387
            s0 += 0.5 * theta * zz - (pb.ldW[kw] + 2.0 * sumlog) / (2.0 * theta);
        }
        __syncthreads();
        if (lane < n) {                                                                    // s_vec  (:389)
            double acc = w.qv[lane];
            #pragma unroll 8
            for (int k = 0; k < n; ++k) acc += w.At[k + ldn * lane] * w.dsv[k];
            #pragma unroll 8
            for (int g = 0; g < m; ++g) acc += w.Lt[g + ldm * lane] * w.hv[g] + w.G[g + ldm * lane] * w.dlv[g];
            w.sv0[lane] = acc;
        }
        prod(m, n, m, [&](int g) { return w.H + g; }, ldm, [&](int j) { return w.Lt + ldm * j; }, 1,
             [&](int g, int j, double c) { w.F[g + ldm * j] = w.G[g + ldm * j] + c; });                           // H L + G  (into F, dead by now)
        __syncthreads();
        // S = Symmetric(Q + A'(D S)A + L'(HL + G) + G'L)  (:390-391): the upper triangle rules
        prod(n, n, n, [&](int i) { return w.At + ldn * i; }, 1, [&](int j) { return w.T + ldn * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) w.U[i + ldn * j] = w.Qs[i + ldn * j] + c; });
        prod(n, n, m, [&](int i) { return w.Lt + ldm * i; }, 1, [&](int j) { return w.F + ldm * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) w.U[i + ldn * j] += c; });
        prod(n, n, m, [&](int i) { return w.G + ldm * i; }, 1, [&](int j) { return w.Lt + ldm * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) { const double v = w.U[i + ldn * j] + c; w.U[i + ldn * j] = v; w.U[j + ldn * i] = v; } });
        if (lane < n) w.sv[lane] = w.sv0[lane];
        __syncthreads();
        ldsd *tmp = w.S; w.S = w.U; w.U = tmp; w.Z = tmp;
        s1 = s0;
        if (dump) {
            if (dump->s && lane == 0) dump->s[t] = s1;
            if (dump->sv && lane < n) dump->sv[(size_t)t * n + lane] = w.sv[lane];
            if (dump->S) each(n, n, [&](int i, int j) { dump->S[(size_t)t * n2 + i + n * j] = w.S[i + ldn * j]; });
        }
    }
    value = s1;
    return 0;
}

#include "wide16.h"
#include "wide32.h"

// simulate_dynamics(problem, x_0, u_array)  (ileqg.jl:18-38) into (xo, uo)
__device__ void rollout_open(const WideProblemDev &pb, Ws &w, const double *x0, const double *u0, double *xo, double *uo) {
    const int n = pb.n, m = pb.m, N = pb.N, lane = threadIdx.x;
    const int ldn = n | 1;
    for (int e = lane; e < n * n; e += 64) w.At[e % n + ldn * (e / n)] = pb.A[e];
    if (lane < n) { w.xt[lane] = x0[lane]; xo[lane] = x0[lane]; }
    __syncthreads();
    for (int t = 0; t < N; ++t) {
        if (lane < m) { const double v = u0[(size_t)t * m + lane]; w.ut[lane] = v; if (uo) uo[(size_t)t * m + lane] = v; }
        __syncthreads();
        double xn = 0.0;
        if (lane < n) {
            double acc = 0.0, accb = 0.0;
            #pragma unroll 8
            for (int j = 0; j < n; ++j) acc += w.At[lane + ldn * j] * w.xt[j];
            #pragma unroll 8
            for (int g = 0; g < m; ++g) accb += w.Bm[lane + ldn * g] * w.ut[g];
            const double xi = w.xt[lane];
            xn = acc + accb + pb.kappa * (xi * xi * xi);
        }
        __syncthreads();
        if (lane < n) { w.xt[lane] = xn; xo[(size_t)(t + 1) * n + lane] = xn; }
    }
    __syncthreads();
}

// simulate_dynamics(problem, x_array, l_array + eps dl, L_array)  (ileqg.jl:62-87, :509-517); returns maximum(norm.(l .- u_new))  (:539)
__device__ double rollout_closed(const WideProblemDev &pb, Ws &w, const double *xbar, const double *l, const double *dl, const double *L,
                                 const double eps, double *xo, double *uo) {
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m, lane = threadIdx.x;
    const int ldn = n | 1;
    for (int e = lane; e < n * n; e += 64) w.At[e % n + ldn * (e / n)] = pb.A[e];
    if (lane < n) { w.xt[lane] = xbar[lane]; xo[lane] = xbar[lane]; }
    double best = -INFINITY;
    bool nan_seen = false;
    __syncthreads();
    for (int t = 0; t < N; ++t) {
        if (lane < n) w.xb[lane] = w.xt[lane] - xbar[(size_t)t * n + lane];
        __syncthreads();
        double diff2 = 0.0;
        if (lane < m) {
            const double *Lt = L + (size_t)t * nm;
            double acc = 0.0;
            #pragma unroll 8
            for (int j = 0; j < n; ++j) acc += Lt[lane + m * j] * w.xb[j];
            const double lt = l[(size_t)t * m + lane];
            const double un = (dl ? lt + eps * dl[(size_t)t * m + lane] : lt) + acc;
            w.ut[lane] = un; uo[(size_t)t * m + lane] = un;
            const double df = lt - un;
            diff2 = df * df;
        }
        const double v = sqrt(wsum(diff2));
        if (!nan_seen) {
            if (v > best || v != v) best = v;          // Julia's maximum propagates NaN
            if (v != v) nan_seen = true;
        }
        __syncthreads();
        double xn = 0.0;
        if (lane < n) {
            double acc = 0.0, accb = 0.0;
            #pragma unroll 8
            for (int j = 0; j < n; ++j) acc += w.At[lane + ldn * j] * w.xt[j];
            #pragma unroll 8
            for (int g = 0; g < m; ++g) accb += w.Bm[lane + ldn * g] * w.ut[g];
            const double xi = w.xt[lane];
            xn = acc + accb + pb.kappa * (xi * xi * xi);
        }
        __syncthreads();
        if (lane < n) { w.xt[lane] = xn; xo[(size_t)(t + 1) * n + lane] = xn; }
    }
    __syncthreads();
    return best;
}

__host__ __device__ inline size_t wide_lds_doubles(const int n, const int m) {            // what carve() hands out
    const size_t ldn = n | 1, ldm = m | 1, sb = ldn * m, sg = ldm * n;
    return 6 * ldn * n + sb + (sb > sg ? sb : sg) + 3 * sg + 2 * ldm * m + (size_t)9 * n + (size_t)6 * m;
}
__device__ inline void carve(Ws &w, ldsd *p, const int n, const int m) {                  // the workgroup's LDS area -> named matrices
    const int ldn = n | 1, ldm = m | 1, sn = ldn * n, sb = ldn * m, sg = ldm * n, sf = sb > sg ? sb : sg, sh = ldm * m;
    w.S = p; p += sn; w.U = p; w.Z = p; p += sn; w.DS = p; p += sn; w.T = p; p += sn; w.At = p; p += sn;      // (Z lives in U: see sweep)
    w.Bm = p; p += sb; w.F = p; p += sf; w.G = p; p += sg; w.Lt = p; p += sg; w.H = p; p += sh;
    w.sv = p; p += n; w.z = p; p += n; w.dsv = p; p += n; w.qv = p; p += n; w.sv0 = p; p += n; w.xt = p; p += n; w.xb = p; p += n;
    w.g = p; p += m; w.dlv = p; p += m; w.rv = p; p += m; w.ut = p; p += m; w.hv = p; p += m;
    w.Qs = p; p += sn; w.Ps = p; p += sg; w.Rs = p; p += sh; w.qvs = p; p += n; w.rvs = p; p += m; w.dA = p; p += n;
}


// NT, MT > 0: every sweep and rollout in registers in the block form of wide32.h (n <= 16 NT, m <= 16 MT; the workgroup's LDS holds only the
// rounds' 0 / 1 tables: four workgroups -- one wavefront per SIMD -- per CU at every size); NT = 0: the general LDS sweeps, or wide16.h's form
template <int NT, int MT>
__global__ __launch_bounds__(64) void wide_solve_kernel(const WideArgs a) {
    extern __shared__ double lds[];
    constexpr bool s32 = NT > 0;
    constexpr int NT_ = s32 ? NT : 1, MT_ = s32 ? MT : 1;
    const WideProblemDev &pb = a.pb;
    const OptsDev &op = a.op;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m, lane = threadIdx.x, b = blockIdx.x;
    Ws w = {};
    if (!s32) {
        carve(w, (ldsd *)lds, n, m);
        for (int e = lane; e < nm; e += 64) w.Bm[e % n + (n | 1) * (e / n)] = pb.B[e];
    }
    const size_t xstr = (size_t)(N + 1) * n, ustr = (size_t)N * m;
    double *const xs = a.xs + (size_t)b * 2 * xstr, *const us = a.us + (size_t)b * 2 * ustr;
    double *const Lg = a.L + (size_t)b * N * nm, *const dlg = a.dl + (size_t)b * N * m;
    const double theta = a.theta[b];
    const bool hist_on = a.hist && b == 0;
    // ---- initialize!  (ileqg.jl:214-236)
    double mu = 0.0, delta = op.delta_0, d_cur = INFINITY, value_cur = INFINITY, eps_init = op.eps_init;
    int iter = 0, n_ls = 0, hn = 0, nom = 0, status = ST_RUNNING;
    for (size_t e = lane; e < (size_t)N * nm; e += 64) Lg[e] = 0.0;                         // :230-232
    Tiles tl = {};
    // n <= 16, m <= 4: the sweeps run in registers on the matrix pipe (wide16.h); the exchange area of its gain solve is T's place in LDS
    const bool s16 = !s32 && a.fast16 && n >= 12 && n <= 16 && m <= 4;
    ldsd *const tab16 = s32 ? (ldsd *)lds : (ldsd *)lds + wide_lds_doubles(n, m);        // (its tables: behind the general kernel's area)
    if (s16 || s32) setup16(pb, tab16);
    // block form, time-invariant cost: the cost gradients of a slot's trajectory are formed behind its rollout, 16 steps per product (grad32)
    double *const gqs = s32 ? a.gq + (size_t)b * 2 * N * n : nullptr, *const grs = s32 ? a.gr + (size_t)b * 2 * N * m : nullptr;
    double *const gcs = s32 ? a.gc + (size_t)b * 2 * 64 : nullptr;
    auto grads = [&](const int slot) {
        if constexpr (s32) {
            if (!pb.cost_tv) grad32<NT_, MT_>(pb, xs + (size_t)slot * xstr, us + (size_t)slot * ustr, gqs + (size_t)slot * N * n, grs + (size_t)slot * N * m, gcs + slot * 64);
        }
    };
    if constexpr (s32) { rollout32<false, NT_, MT_>(pb, a.x0, a.u0, nullptr, nullptr, 0.0, xs, us); grads(0); }
    else if (s16) rollout16<false>(pb, a.x0, a.u0, nullptr, nullptr, 0.0, xs, us);          // :225, :228
    else rollout_open(pb, w, a.x0, a.u0, xs, us);
    auto run_sweep = [&](const double *xt, const double *ut, const double mu_, auto gain_c, auto zero_c, double &val) -> int {
        constexpr bool gain = decltype(gain_c)::value, zeroL = decltype(zero_c)::value;
        if constexpr (s32) {
            const int slot = (xt == xs) ? 0 : 1;
            return sweep32<gain, zeroL, NT_, MT_>(pb, tab16, xt, ut, theta, mu_, Lg, dlg, val, gqs + (size_t)slot * N * n, grs + (size_t)slot * N * m, gcs + slot * 64);
        }
        else {
            if (s16) return sweep16<gain, zeroL>(pb, w.T, tab16, xt, ut, theta, mu_, Lg, dlg, val);
            tl.x = xt; tl.u = ut;
            return sweep(pb, w, tl, theta, mu_, gain, zeroL, Lg, dlg, nullptr, nullptr, val);
        }
    };
    const std::integral_constant<bool, true> yes;
    const std::integral_constant<bool, false> no;
    {
        const int rc = run_sweep(xs, us, mu, no, yes, value_cur);                          // :233-235
        if (rc) status = 1;
    }
    // ---- while true: step!  (:640-654)
    while (status == ST_RUNNING) {
        iter++;                                                                             // :599
        double *xn = xs + (size_t)nom * xstr, *un = us + (size_t)nom * ustr;                // x_array, l_array
        double *xc = xs + (size_t)(nom ^ 1) * xstr, *uc = us + (size_t)(nom ^ 1) * ustr;    // the candidate's
        int restarts = 0;
        for (;;) {                                                                          // solve_approximate_dp!  (:359-403)
            double dummy;
            const int rc = run_sweep(xn, un, mu, yes, no, dummy);
            if (rc == 0) break;
            if (rc == 2) { status = 2; break; }
            delta = fmax(op.delta_0, delta * op.delta_0);                                   // increase_mu_and_delta!  (:471-474)
            mu = fmax(op.mu_min, mu * delta);
            if (++restarts > 400 || !isfinite(mu)) { status = 5; break; }                   // (the reference would spin)
        }
        if (status != ST_RUNNING) break;
        // line_search!  (:494-592)
        const double cur = value_cur;
        double eps = eps_init;
        int count = 0;
        for (;;) {
            count++;                                                                        // :505
            if (count > 4000) { status = 7; break; }                                        // (App. B.5)
            n_ls++;
            double d_new;                                                                   // :509-517
            if constexpr (s32) { d_new = rollout32<true, NT_, MT_>(pb, xn, un, dlg, Lg, eps, xc, uc); grads(nom ^ 1); }
            else d_new = s16 ? rollout16<true>(pb, xn, un, dlg, Lg, eps, xc, uc) : rollout_closed(pb, w, xn, un, dlg, Lg, eps, xc, uc);
            double newv;
            const int rc = run_sweep(xc, uc, mu, no, no, newv);                      // :520-528
            if (rc) { eps *= op.lambda; continue; }                                         // :529-535
            if (hist_on) {                                                                  // :537
                if (hn < a.hist_cap && lane == 0) { a.hist[2 * (size_t)hn] = eps; a.hist[2 * (size_t)hn + 1] = newv - cur; }
                hn++;
            }
            if (!(isapprox_default(newv, cur) || newv < cur)) {                             // :538
                eps *= op.lambda;                                                           // :557
                if (!(eps < op.eps_min)) continue;                                          // :558
            }
            d_cur = d_new; value_cur = newv; nom ^= 1;                                      // :539-555 / :559-575
            break;
        }
        if (status != ST_RUNNING) break;
        if (op.adaptive) {                                                                  // :582-591
            if (count == 1) eps_init = fmin(op.eps_init, eps / op.lambda);
            else { while (eps < op.eps_min) eps = eps / op.lambda; eps_init = eps; }
        }
        if (op.d > d_cur && mu <= op.mu_min) status = 0;                                    // :642
        else if (iter == op.iter_max) status = 3;                                           // :648
    }
    if (lane == 0) {
        const bool ok = status == 0 || status == 3;
        const double val = ok ? value_cur : INFINITY;                                       // catch -> Inf  (cross_entropy...jl:163)
        a.nom[b] = nom;
        if (a.out_value) a.out_value[b] = val;
        if (a.out_status) a.out_status[b] = status;
        if (a.out_iters) a.out_iters[b] = iter;
        if (a.out_ls) a.out_ls[b] = n_ls;
        if (a.out_cost) a.out_cost[b] = ok ? val + a.kl_bound / theta : INFINITY;           // :193
        if (hist_on && a.hist_n) a.hist_n[0] = hn;
    }
}


// ---- operator forms ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u01(unsigned hi, unsigned lo) {       // 53-bit uniform in [0, 1)
    return (double)((((unsigned long long)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// c(k, x_t, u_t) of the LQ family with the gradients left in w.qv / w.rv  (x_t in w.xt, u_t in w.ut)
__device__ double stage_cost(const WideProblemDev &pb, Ws &w, const int t) {
    const int n = pb.n, m = pb.m, lane = threadIdx.x, kc = pb.cost_tv ? t : 0;
    const double *Qk = pb.Q + (size_t)kc * n * n, *Rk = pb.R + (size_t)kc * m * m, *Pk = pb.P + (size_t)kc * n * m;
    const double *qvk = pb.qv + (size_t)kc * n, *rvk = pb.rv + (size_t)kc * m;
    double part = 0.0;
    if (lane < n) {
        double qx = 0.0, pu = 0.0;
        #pragma unroll 8
        for (int j = 0; j < n; ++j) qx += Qk[lane + n * j] * w.xt[j];
        #pragma unroll 8
        for (int g = 0; g < m; ++g) pu += Pk[g + m * lane] * w.ut[g];
        w.qv[lane] = qx + pu + qvk[lane];
        part = w.xt[lane] * (0.5 * qx + qvk[lane]);
    } else if (lane >= 32 && lane - 32 < m) {
        const int g = lane - 32;
        double ru = 0.0, px = 0.0;
        #pragma unroll 8
        for (int g2 = 0; g2 < m; ++g2) ru += Rk[g + m * g2] * w.ut[g2];
        #pragma unroll 8
        for (int j = 0; j < n; ++j) px += Pk[g + m * j] * w.xt[j];
        w.rv[g] = ru + px + rvk[g];
        part = w.ut[g] * (0.5 * ru + px + rvk[g]);
    }
    return wsum(part) + pb.q0[kc];
}
// h(x_N) with its gradient left in w.qv  (x_N in w.xt)
__device__ double terminal_cost(const WideProblemDev &pb, Ws &w) {
    const int n = pb.n, lane = threadIdx.x;
    double part = 0.0;
    if (lane < n) {
        double acc = 0.0;
        #pragma unroll 8
        for (int j = 0; j < n; ++j) acc += pb.Qf[lane + n * j] * w.xt[j];
        w.qv[lane] = acc + pb.qvf[lane];
        part = w.xt[lane] * (0.5 * acc + pb.qvf[lane]);
    }
    return wsum(part) + pb.q0f;
}

#if WIDE_PART & 1
__global__ __launch_bounds__(64) void wide_op_kernel(const WideOpArgs a) {
    extern __shared__ double lds[];
    const WideProblemDev &pb = a.pb;
    const int n = pb.n, m = pb.m, N = pb.N, n2 = n * n, nm = n * m, mm = m * m, lane = threadIdx.x, ldn = n | 1;
    const long b = blockIdx.x;
    Ws w;
    carve(w, (ldsd *)lds, n, m);
    for (int e = lane; e < nm; e += 64) w.Bm[e % n + ldn * (e / n)] = pb.B[e];
    __syncthreads();
    if (a.opcode == WOP_ROLL_OPEN) {                                  // simulate_dynamics(problem, x_0, u_array)  ileqg.jl:18-38
        rollout_open(pb, w, a.x0, a.u, a.x_out, nullptr);
    } else if (a.opcode == WOP_ROLL_FEEDBACK) {                       // simulate_dynamics(problem, x_array, l_array, L_array)  :62-87
        rollout_closed(pb, w, a.xbar, a.l, nullptr, a.L, 0.0, a.x_out, a.u_out);
    } else if (a.opcode == WOP_COST || a.opcode == WOP_APPROX) {      // integrate_cost :115-124 / approximate_model :258-322
        double total = 0.0;
        for (int t = 0; t <= N; ++t) {
            if (lane < n) w.xt[lane] = a.xbar[(size_t)t * n + lane];
            if (t < N && lane < m) w.ut[lane] = a.u[(size_t)t * m + lane];
            __syncthreads();
            const double c = (t < N) ? stage_cost(pb, w, t) : terminal_cost(pb, w);
            total += c;
            __syncthreads();
            if (a.opcode == WOP_APPROX) {
                const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
                if (lane == 0) a.q[t] = c;
                if (lane < n) a.qv[(size_t)t * n + lane] = w.qv[lane];
                const double *Qs = (t < N) ? pb.Q + (size_t)kc * n2 : pb.Qf;
                for (int e = lane; e < n2; e += 64) a.Q[(size_t)t * n2 + e] = Qs[e];
                if (t < N) {
                    if (lane < m) a.r[(size_t)t * m + lane] = w.rv[lane];
                    for (int e = lane; e < mm; e += 64) a.R[(size_t)t * mm + e] = pb.R[(size_t)kc * mm + e];
                    for (int e = lane; e < nm; e += 64) { a.P[(size_t)t * nm + e] = pb.P[(size_t)kc * nm + e]; a.B[(size_t)t * nm + e] = pb.B[e]; }
                    for (int e = lane; e < n2; e += 64) {
                        const int i = e % n, j = e / n;
                        a.A[(size_t)t * n2 + e] = pb.A[e] + ((i == j) ? 3.0 * pb.kappa * (w.xt[i] * w.xt[i]) : 0.0);
                        a.W[(size_t)t * n2 + e] = pb.W[(size_t)kw * n2 + e];
                    }
                }
            }
            __syncthreads();
        }
        if (a.opcode == WOP_COST && lane == 0) a.cost_out[0] = total;
    } else if (a.opcode == WOP_NOISY) {                               // simulate_dynamics(..., rng)  :44-55 / :94-109 + integrate_cost
        double *xo = a.x_out ? a.x_out + (size_t)b * (N + 1) * n : nullptr, *uo = a.u_out ? a.u_out + (size_t)b * N * m : nullptr;
        for (int e = lane; e < n2; e += 64) w.At[e % n + ldn * (e / n)] = pb.A[e];
        if (lane < n) w.xt[lane] = a.xbar[lane];
        double total = 0.0, znext = 0.0;
        __syncthreads();
        for (int t = 0; t < N; ++t) {
            const int kw = pb.W_tv ? t : 0;
            if (xo && lane < n) xo[(size_t)t * n + lane] = w.xt[lane];
            if (a.L && lane < n) w.xb[lane] = w.xt[lane] - a.xbar[(size_t)t * n + lane];
            double zt = 0.0;
            if (lane < n) {
                if (a.z) zt = a.z[((size_t)b * N + t) * n + lane];
                else if ((t & 1) == 0) {                              // both outputs of one Box-Muller transform: steps t and t + 1
                    unsigned r[4];
                    philox4x32_10((unsigned)b, (unsigned)((unsigned long long)b >> 32), (unsigned)(t >> 1), (unsigned)lane, (unsigned)a.seed,
                                  (unsigned)(a.seed >> 32), r);
                    ratn_box_muller(u01(r[0], r[1]), u01(r[2], r[3]), &zt, &znext);
                } else zt = znext;
                w.z[lane] = zt;
            }
            __syncthreads();
            if (lane < m) {
                double un = a.l[(size_t)t * m + lane];
                if (a.L) {
                    const double *Lt = a.L + (size_t)t * nm;
                    double acc = 0.0;
                    #pragma unroll 8
                    for (int j = 0; j < n; ++j) acc += Lt[lane + m * j] * w.xb[j];
                    un += acc;
                }
                w.ut[lane] = un;
                if (uo) uo[(size_t)t * m + lane] = un;
            }
            __syncthreads();
            total += stage_cost(pb, w, t);
            double xn = 0.0;
            if (lane < n) {
                double acc = 0.0, accb = 0.0, wn = 0.0;
                #pragma unroll 8
                for (int j = 0; j < n; ++j) acc += w.At[lane + ldn * j] * w.xt[j];
                #pragma unroll 8
                for (int g = 0; g < m; ++g) accb += w.Bm[lane + ldn * g] * w.ut[g];
                const double *Lw = a.Wchol + (size_t)kw * n2;
                for (int j = 0; j <= lane; ++j) wn += Lw[lane + n * j] * w.z[j];          // chol_lower(W(k)) z
                const double xi = w.xt[lane];
                xn = (acc + accb + pb.kappa * (xi * xi * xi)) + wn;
            }
            __syncthreads();
            if (lane < n) w.xt[lane] = xn;
            __syncthreads();
        }
        if (xo && lane < n) xo[(size_t)N * n + lane] = w.xt[lane];
        total += terminal_cost(pb, w);
        if (a.cost_out && lane == 0) a.cost_out[b] = total;
    } else if (a.opcode == WOP_DP_GAIN || a.opcode == WOP_DP_EVAL) {  // solve_approximate_dp! :341-406 / solve_approximate_dp :412-465
        Tiles tl = {};
        tl.q = a.q + (size_t)b * (N + 1); tl.qv = a.qv + (size_t)b * (N + 1) * n; tl.Q = a.Q + (size_t)b * (N + 1) * n2;
        tl.r = a.r + (size_t)b * N * m; tl.R = a.R + (size_t)b * N * mm; tl.P = a.P + (size_t)b * N * nm;
        tl.A = a.A + (size_t)b * N * n2; tl.B = a.B + (size_t)b * N * nm;
        Dump dp = {a.ds, a.dsv, a.dS, a.dg, a.dG, a.dH};
        const Dump *dump = (b == 0 && (a.ds || a.dsv || a.dS || a.dg || a.dG || a.dH)) ? &dp : nullptr;
        double *Lg = a.Lio + (size_t)b * N * nm;
        const double theta = a.theta[b];
        double value = INFINITY;
        int status = 0;
        if (a.opcode == WOP_DP_GAIN) {
            double mu = a.mu[b], delta = a.delta[b];
            int restarts = 0;
            for (;;) {
                const int rc = sweep(pb, w, tl, theta, mu, true, false, Lg, a.dl_out + (size_t)b * N * m, nullptr, dump, value);
                if (rc == 0) break;
                if (rc == 2) { status = 2; break; }
                delta = fmax(a.op.delta_0, delta * a.op.delta_0);                           // increase_mu_and_delta!  (:471-474)
                mu = fmax(a.op.mu_min, mu * delta);
                if (++restarts > 400 || !isfinite(mu)) { status = 5; break; }
            }
            if (lane == 0) { a.mu[b] = mu; a.delta[b] = delta; }
        } else {
            const int rc = sweep(pb, w, tl, theta, a.mu_in[b], false, false, Lg, nullptr, a.dlin ? a.dlin + (size_t)b * N * m : nullptr, dump, value);
            if (rc) status = 2;
        }
        if (lane == 0) {
            if (a.status) a.status[b] = status;
            if (a.value) a.value[b] = status ? INFINITY : value;
        }
    }
}

#endif  // WIDE_PART & 1
}  // namespace

#if WIDE_PART & 1
size_t wide_lds_bytes(int n, int m) { return sizeof(double) * wide_lds_doubles(n, m); }
#endif


// The file is compiled in two parts (Makefile: WIDE_PART): 1 = the general kernel, wide16.h's form and the operator kernels; 2 = the block-form
// kernels of wide32.h -- their own unit because they are built WITHOUT -amdgpu-mfma-vgpr-form (with it clang 22's "Rewrite AGPR-Copy-MFMA" pass
// crashes on kernels that spill, and the <2, 2> instantiation does).
hipError_t launch_wide32_solve(const WideArgs &a, hipStream_t s);
#if WIDE_PART & 2
hipError_t launch_wide32_solve(const WideArgs &a, hipStream_t s) {       // the LDS holds only the rounds' 0 / 1 tables
    const size_t lds32 = sizeof(double) * W16_LDS;
    if (a.pb.img_nt == 1 && a.pb.img_mt == 1) hipLaunchKernelGGL((wide_solve_kernel<1, 1>), dim3(a.B), dim3(64), lds32, s, a);
    else if (a.pb.img_mt == 1) hipLaunchKernelGGL((wide_solve_kernel<2, 1>), dim3(a.B), dim3(64), lds32, s, a);
    else hipLaunchKernelGGL((wide_solve_kernel<2, 2>), dim3(a.B), dim3(64), lds32, s, a);
    return hipGetLastError();
}
#endif
#if WIDE_PART & 1
bool wide32_applies(const WideArgs &a) {
    const bool s16 = a.fast16 && a.pb.n >= 12 && a.pb.n <= 16 && a.pb.m <= 4;
    return a.fast32 && !s16;
}
hipError_t launch_wide_solve(const WideArgs &a, hipStream_t s) {
    const bool s16 = a.fast16 && a.pb.n >= 12 && a.pb.n <= 16 && a.pb.m <= 4;
    if (wide32_applies(a)) return launch_wide32_solve(a, s);            // every other size: the block form in registers (wide32.h)
    const size_t lds = wide_lds_bytes(a.pb.n, a.pb.m) + (s16 ? sizeof(double) * W16_LDS : 0);
    if (lds > 64 * 1024) {               // (per device: set on every launch that needs it, the call is cheap)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_solve_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((wide_solve_kernel<0, 0>), dim3(a.B), dim3(64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_wide_op(const WideOpArgs &a, hipStream_t s) {
    const size_t lds = wide_lds_bytes(a.pb.n, a.pb.m);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_op_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wide_op_kernel, dim3((unsigned)a.count), dim3(64), lds, s, a);
    return hipGetLastError();
}
#endif  // WIDE_PART & 1
