// wide.hip -- the complete solve! of one theta-sample for problems beyond the 12 + 4 tile of kernels.hip (n <= 32, m <= 32, LQ family).
//
// The MFMA kernels of kernels.hip are specialised for n <= 12, m <= 4 (every BASELINE configuration): there [A|B] is one 12 x 16 operand
// and the value function one accumulator tile.  The reference itself takes its dimensions from the arrays (ileqg.jl:229); this kernel
// covers what lies beyond the tile with ONE WORKGROUP (one wavefront) PER SAMPLE and every matrix of a backward step in LDS at its own
// size (column-major with odd leading dimensions, 6 n^2 + 4 n m + 2 m^2 doubles + padding: 104 KB at n = m = 32 of the CU's 160 KB).  The whole solve! runs inside the
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
// The arithmetic is plain FP64 vector code: this is the general-size path, not the measured one.
#include <hip/hip_runtime.h>
#include <math.h>

#include "wide.h"
#include "device_utils.h"

namespace {

struct Ws {
    double *S, *U, *Z, *DS, *T, *At, *Bm, *F, *G, *Lt, *H, *Hc;
    double *sv, *z, *dsv, *qv, *sv0, *xt, *xb, *g, *dlv, *rv, *ut, *hv;
    double *idle;          // [WIDE_MAX]: what lanes without a column of their own READ in the lockstep factorisation (they store nothing)
                           // Invariant of U / Hc: chol_fwd leaves the factor in the UPPER triangle; its matrix lanes also overwrite the
                           // strictly lower triangle of their column with intermediate values, so every consumer (back_all, the
                           // products forming D S) reads entries (i, j) with i <= j only
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
// C(i, j) = sum_k A(i, k) B(k, j), A(i, k) = pa(i)[k * sa], B(k, j) = pb(j)[k * sb]; store(i, j, c) for every entry
template <class PA, class PB, class ST>
__device__ __forceinline__ void prod(const int rows, const int cols, const int K, PA pa, const int sa, PB pb, const int sb, ST store) {
    const int li = threadIdx.x & 7, lj = threadIdx.x >> 3;
    for (int j0 = lj; j0 < cols; j0 += 16) {
        const int j1 = j0 + 8;
        const bool vj = j1 < cols;
        const double *b0 = pb(j0), *b1 = pb(vj ? j1 : j0);
        for (int i0 = li; i0 < rows; i0 += 16) {
            const int i1 = i0 + 8;
            const bool vi = i1 < rows;
            const double *a0 = pa(i0), *a1 = pa(vi ? i1 : i0);
            double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
            #pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const double x0 = a0[k * sa], x1 = a1[k * sa], y0 = b0[k * sb], y1 = b1[k * sb];
                c00 += x0 * y0; c01 += x0 * y1; c10 += x1 * y0; c11 += x1 * y1;
            }
            store(i0, j0, c00);
            if (vj) store(i0, j1, c01);
            if (vi) store(i1, j0, c10);
            if (vi && vj) store(i1, j1, c11);
        }
    }
}

// Cholesky factorisation X = U'U in place (upper triangle of the k x k matrix Um, k <= 32) FUSED with the forward substitutions
// U'Y = R for nrhs <= 32 right-hand-side columns (Rm, in place) and one more right-hand side `vec`, in lockstep over the rows: lane c < k
// owns column c of the matrix, lane 32 + c column c of R (every lane runs the same dot-product loop against column i of U, which is
// final by the time row i is reached), and the extra right-hand side is reduced across the lanes (lane q holds y_q).  Left-looking, so
// each entry is M_ij - sum_k U_ki U_kj in ascending k like the textbook loop.  false <=> a pivot <= 0 or NaN: what LAPACK potrf reports
// and isposdef tests.  yv: lane i < k ends with y_i of the extra right-hand side; sumlog = sum_i log U_ii.
__device__ bool chol_fwd(double *Um, const int k, const int ld, double *Rm, const int nrhs, const int ldr, const double *vec, double *idle,
                         double &yv, double &sumlog) {
    const int lane = threadIdx.x;
    const bool isM = lane < k, isR = lane >= 32 && lane - 32 < nrhs;
    double *col = isM ? Um + (size_t)ld * lane : (isR ? Rm + (size_t)ldr * (lane - 32) : idle);
    double myr = 1.0;
    yv = 0.0;
    for (int i = 0; i < k; ++i) {
        const double *ui = Um + (size_t)ld * i;
        double acc = col[i];
        #pragma unroll 8
        for (int q = 0; q < i; ++q) acc -= ui[q] * col[q];
        const double dot = wsum((lane < i) ? ui[lane] * yv : 0.0);
        const double d = readlane_f64(acc, i);
        if (!(d > 0.0)) return false;
        const double r = sqrt(d), ri = fast_rcp(r);
        // (lanes without a column run the same loop on the shared `idle` vector -- uninitialised, possibly NaN, never part of a result --
        //  and do not write it: no two lanes ever store to one address)
        if (isM || isR) col[i] = (isM && lane == i) ? r : acc * ri;
        if (lane == i) { yv = (vec[i] - dot) * ri; myr = r; }
        __syncthreads();
    }
    sumlog = wsum(isM ? log(myr) : 0.0);
    return true;
}
// back substitutions U X = Y for the nrhs columns of Rm (in place) and the extra right-hand side held in yv (lane i: y_i -> x_i)
__device__ void back_all(const double *Um, const int k, const int ld, double *Rm, const int nrhs, const int ldr, double *idle, double &yv) {
    const int lane = threadIdx.x;
    const bool isR = lane >= 32 && lane - 32 < nrhs;
    double *col = isR ? Rm + (size_t)ldr * (lane - 32) : idle;
    for (int i = k - 1; i >= 0; --i) {
        double acc = col[i];
        #pragma unroll 8
        for (int q = i + 1; q < k; ++q) acc -= Um[i + (size_t)ld * q] * col[q];
        const double dot = wsum((lane > i && lane < k) ? Um[i + (size_t)ld * lane] * yv : 0.0);
        const double ri = fast_rcp(Um[i + (size_t)ld * i]);
        if (isR) col[i] = acc * ri;                       // (idle lanes read the shared idle vector and store nothing)
        if (lane == i) yv = (yv - dot) * ri;
        __syncthreads();
    }
}

// Where a step's quadratic model comes from: the solver forms it on the fly from the trajectory (x, u) and the problem tables; the
// operator forms (rat_dp_*) read caller-built ApproximationResult arrays in the C ABI layout (A != nullptr).
struct Tiles {
    const double *x, *u;
    const double *q, *qv, *Q, *r, *R, *P, *A, *B;
};
struct Dump { double *s, *sv, *S, *g, *G, *H; };        // DynamicProgrammingResult arrays of the operator forms (any may be null)

// solve_approximate_dp (gain = false, :412-465) / one pass of solve_approximate_dp! (gain = true, :341-406) over the trajectory.
// Returns 0, 2 (M not positive definite) or -1 (H not positive definite: the caller raises mu and restarts).
__device__ int sweep(const WideProblemDev &pb, Ws &w, const Tiles &tl, const double theta, const double mu,
                     const bool gain, const bool zeroL, double *Lg, double *dlg, const double *dlin, const Dump *dump, double &value) {
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
    for (int t = N - 1; t >= 0; --t) {
        const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
        const double *Qk = arr ? tl.Q + (size_t)t * n2 : pb.Q + (size_t)kc * n2, *Rk = arr ? tl.R + (size_t)t * mm : pb.R + (size_t)kc * mm;
        const double *Pk = arr ? tl.P + (size_t)t * nm : pb.P + (size_t)kc * nm;
        const double *Wk = pb.W + (size_t)kw * n2, *Wik = pb.Winv + (size_t)kw * n2;
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
        } else {                                                                           // approximate_model at (x_t, u_t)  (:294-313)
            const double *qvk = pb.qv + (size_t)kc * n, *rvk = pb.rv + (size_t)kc * m;
            if (lane < n) w.xt[lane] = x[(size_t)t * n + lane];
            if (lane >= 32 && lane - 32 < m) w.ut[lane - 32] = u[(size_t)t * m + lane - 32];
            __syncthreads();
            each(n, n, [&](int i, int j) { w.At[i + ldn * j] = pb.A[i + n * j] + ((i == j) ? 3.0 * pb.kappa * (w.xt[i] * w.xt[i]) : 0.0); });
            part = 0.0;
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
            q = wsum(part) + pb.q0[kc];
        }
        // M = inv(W) - theta S  (:365) into U;  Z <- S (right-hand sides of the forward substitution)
        each(n, n, [&](int i, int j) { const double sij = w.S[i + ldn * j]; w.U[i + ldn * j] = Wik[i + n * j] - theta * sij; w.Z[i + ldn * j] = sij; });
        __syncthreads();
        double sumlog = 0.0, zreg = 0.0;
        if (!chol_fwd(w.U, n, ldn, w.Z, n, ldn, w.sv, w.idle, zreg, sumlog)) return 2;       // @assert isposdef(M)  :366 / :440;  [Z | z] = U^-T [S | s_vec]
        if (theta == 0.0) {                          // D = I exactly, whatever the size of S: the products below must not see Z'Z
            each(n, n, [&](int i, int j) { w.Z[i + ldn * j] = 0.0; });
            zreg = 0.0;
        }
        if (lane < n) w.z[lane] = zreg;
        __syncthreads();
        prod(n, n, n, [&](int i) { return w.Z + (size_t)ldn * i; }, 1, [&](int j) { return w.Z + (size_t)ldn * j; }, 1,
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
             [&](int j) { return (j < n) ? w.At + (size_t)ldn * j : w.Bm + (size_t)ldn * (j - n); }, 1,
             [&](int i, int j, double c) { if (j < n) w.T[i + ldn * j] = c; else w.F[i + ldn * (j - n)] = c; });   // T = (D S) A,  F = (D S) B
        __syncthreads();
        prod(m, n, n, [&](int g) { return w.Bm + (size_t)ldn * g; }, 1, [&](int j) { return w.T + (size_t)ldn * j; }, 1,
             [&](int g, int j, double c) { w.G[g + ldm * j] = Pk[g + m * j] + c; });                              // G = P + B'(D S) A  (:369)
        prod(m, m, n, [&](int g) { return w.Bm + (size_t)ldn * g; }, 1, [&](int g2) { return w.F + (size_t)ldn * g2; }, 1,
             [&](int g, int g2, double c) {                                                                       // H = Symmetric(R + B'(D S) B + mu I)  (:370-371)
                 if (g <= g2) {
                     const double v = Rk[g + m * g2] + c + ((g == g2) ? mu : 0.0);
                     w.H[g + ldm * g2] = v; w.H[g2 + ldm * g] = v; w.Hc[g + ldm * g2] = v;
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
            if (!chol_fwd(w.Hc, m, ldm, w.Lt, n, ldm, w.hv, w.idle, dreg, dummy)) return -1; // !isposdef(H)  :372
            back_all(w.Hc, m, ldm, w.Lt, n, ldm, w.idle, dreg);
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
        prod(m, n, m, [&](int g) { return w.H + g; }, ldm, [&](int j) { return w.Lt + (size_t)ldm * j; }, 1,
             [&](int g, int j, double c) { w.F[g + ldm * j] = w.G[g + ldm * j] + c; });                           // H L + G  (into F, dead by now)
        __syncthreads();
        // S = Symmetric(Q + A'(D S)A + L'(HL + G) + G'L)  (:390-391): the upper triangle rules
        prod(n, n, n, [&](int i) { return w.At + (size_t)ldn * i; }, 1, [&](int j) { return w.T + (size_t)ldn * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) w.U[i + ldn * j] = Qk[i + n * j] + c; });
        prod(n, n, m, [&](int i) { return w.Lt + (size_t)ldm * i; }, 1, [&](int j) { return w.F + (size_t)ldm * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) w.U[i + ldn * j] += c; });
        prod(n, n, m, [&](int i) { return w.G + (size_t)ldm * i; }, 1, [&](int j) { return w.Lt + (size_t)ldm * j; }, 1,
             [&](int i, int j, double c) { if (i <= j) { const double v = w.U[i + ldn * j] + c; w.U[i + ldn * j] = v; w.U[j + ldn * i] = v; } });
        if (lane < n) w.sv[lane] = w.sv0[lane];
        __syncthreads();
        double *tmp = w.S; w.S = w.U; w.U = tmp;
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

__device__ inline void carve(Ws &w, double *p, const int n, const int m) {                  // the workgroup's LDS area -> named matrices
    const int ldn = n | 1, ldm = m | 1, sn = ldn * n, sb = ldn * m, sg = ldm * n, sf = sb > sg ? sb : sg, sh = ldm * m;
    w.S = p; p += sn; w.U = p; p += sn; w.Z = p; p += sn; w.DS = p; p += sn; w.T = p; p += sn; w.At = p; p += sn;
    w.Bm = p; p += sb; w.F = p; p += sf; w.G = p; p += sg; w.Lt = p; p += sg; w.H = p; p += sh; w.Hc = p; p += sh;
    w.sv = p; p += n; w.z = p; p += n; w.dsv = p; p += n; w.qv = p; p += n; w.sv0 = p; p += n; w.xt = p; p += n; w.xb = p; p += n;
    w.g = p; p += m; w.dlv = p; p += m; w.rv = p; p += m; w.ut = p; p += m; w.hv = p; p += m;
    w.idle = p;
}


__global__ __launch_bounds__(64) void wide_solve_kernel(const WideArgs a) {
    extern __shared__ double lds[];
    const WideProblemDev &pb = a.pb;
    const OptsDev &op = a.op;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m, lane = threadIdx.x, b = blockIdx.x;
    Ws w;
    carve(w, lds, n, m);
    for (int e = lane; e < nm; e += 64) w.Bm[e % n + (n | 1) * (e / n)] = pb.B[e];
    const size_t xstr = (size_t)(N + 1) * n, ustr = (size_t)N * m;
    double *const xs = a.xs + (size_t)b * 2 * xstr, *const us = a.us + (size_t)b * 2 * ustr;
    double *const Lg = a.L + (size_t)b * N * nm, *const dlg = a.dl + (size_t)b * N * m;
    const double theta = a.theta[b];
    const bool hist_on = a.hist && b == 0;
    // ---- initialize!  (ileqg.jl:214-236)
    double mu = 0.0, delta = op.delta_0, d_cur = INFINITY, value_cur = INFINITY, eps_init = op.eps_init;
    int iter = 0, n_ls = 0, hn = 0, nom = 0, status = ST_RUNNING;
    for (size_t e = lane; e < (size_t)N * nm; e += 64) Lg[e] = 0.0;                         // :230-232
    rollout_open(pb, w, a.x0, a.u0, xs, us);                                                // :225, :228
    Tiles tl = {};
    {
        tl.x = xs; tl.u = us;
        const int rc = sweep(pb, w, tl, theta, mu, false, true, Lg, dlg, nullptr, nullptr, value_cur);     // :233-235
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
            tl.x = xn; tl.u = un;
            const int rc = sweep(pb, w, tl, theta, mu, true, false, Lg, dlg, nullptr, nullptr, dummy);
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
            const double d_new = rollout_closed(pb, w, xn, un, dlg, Lg, eps, xc, uc);       // :509-517
            double newv;
            tl.x = xc; tl.u = uc;
            const int rc = sweep(pb, w, tl, theta, mu, false, false, Lg, dlg, nullptr, nullptr, newv);     // :520-528
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

__global__ __launch_bounds__(64) void wide_op_kernel(const WideOpArgs a) {
    extern __shared__ double lds[];
    const WideProblemDev &pb = a.pb;
    const int n = pb.n, m = pb.m, N = pb.N, n2 = n * n, nm = n * m, mm = m * m, lane = threadIdx.x, ldn = n | 1;
    const long b = blockIdx.x;
    Ws w;
    carve(w, lds, n, m);
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
                    const double rad = sqrt(-2.0 * log(1.0 - u01(r[0], r[1])));
                    double sn, cs;
                    sincos(6.283185307179586476925286766559 * u01(r[2], r[3]), &sn, &cs);
                    zt = rad * cs; znext = rad * sn;
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

}  // namespace

size_t wide_lds_bytes(int n, int m) {
    const size_t ldn = n | 1, ldm = m | 1, sb = ldn * m, sg = ldm * n;
    return sizeof(double) * (6 * ldn * n + sb + (sb > sg ? sb : sg) + 2 * sg + 2 * ldm * m + (size_t)7 * n + (size_t)5 * m + 64);
}

hipError_t launch_wide_solve(const WideArgs &a, hipStream_t s) {
    const size_t lds = wide_lds_bytes(a.pb.n, a.pb.m);
    if (lds > 64 * 1024) {               // (per device: set on every launch that needs it, the call is cheap)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(wide_solve_kernel, dim3(a.B), dim3(64), lds, s, a);
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
