// wide32.h -- the backward sweep and the rollouts of wide.hip for every size beyond wide16.h's (n <= 32, m <= 32), in registers on the matrix
// pipe: wide16.h's formulation on BLOCKS of 16 x 16 tiles.  Included by wide.hip (inside its anonymous namespace, behind wide16.h).
//
// A matrix is a Blk<RT, CT>: RT x CT tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (d4: register r of lane (g, j) = element
// (4 r + g, j) of the tile); the n states take NT = 1 or 2 tiles, the m controls MT = 1 or 2, the affine parts (s_vec, q_vec, g, dl) are
// one-column blocks (column 0 of a tile).  The one product primitive is pmm(A, B, C): C += A'B = sum over the row tiles k and the four
// K-slices s of MFMA(A[k][i][s], B[k][j][s]) -- register s of a tile is the A operand of slice s of (tile)'(.) and the B operand of slice
// s of (.)'(tile), so a step chains through registers without moving data between lanes; symmetric left factors (S, M^-1, H, H^-1, Q, R)
// are their own transposes.
//   M = inv(W) - theta S            -M^-1 by the symmetric sweep with 2 x 2 block pivots over the blocks (8 NT rounds, NT^2 MFMAs each:
//                                   M' = M o mask - t_a'(Bk t_b) per tile (a, b), t_a = the pivot rows' part in column tile a);
//                                   isposdef(M) <=> all leading minors > 0; logdet from the running product of the block determinants      ileqg.jl:365-366
//   X = S [A | B | .] + [0 | 0 | s_vec]     Y = theta M^-1 X     T = X + S Y = (D S)[A | B | S^-1 s_vec]                                      :367
//   F11 = Q + A'T_A,  F1a = q_vec + A'T_a,  G = P + B'T_A,  H = R + mu I + B'T_B,  g = r + B'T_a                                              :368-371
//   gain sweep: -H^-1 by the same elimination (8 MT rounds; leading minors > 0 <=> isposdef(H), else the caller raises mu), L = -H^-1 G,
//   dl = -H^-1 g                                                                                                                            :372-382
//   S' = F11 + L'(H L + G) + G'L,  s_vec' = F1a + L'(H dl + g) + G'dl;  scalars accumulate per lane                                          :383-391
// The step's cost gradients (approximate_model, :294-313) are one-column products on [x_t; u_t].  Problem tables are re-read from L2 where
// a step uses them (a block of four tiles is 32 registers: the step keeps ~40 tiles live at its widest), the next step's (x, u, L) are
// fetched while this one runs.  n = m = 32: ~740 MFMAs + ~2 k vector instructions per step, ~60 k cycles, ONE wavefront per SIMD -- against
// 173 k cycles per step at one wavefront per CU (112 KB of LDS) in the general sweep().
#pragma once

template <int RT, int CT> struct Blk { d4 t[RT][CT]; };
// Phase fence for the instruction scheduler: without it every table load of a step is hoisted to the step's top (a block of four tiles is
// 32 registers; a dozen blocks in flight spill), with it a block is fetched where it is used.
#define W32_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int RT, int CT>
__device__ __forceinline__ void blk_zero(Blk<RT, CT> &o) {
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) o.t[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
}
// C += A'B
template <int KT, int IT, int JT>
__device__ __forceinline__ void pmm(const Blk<KT, IT> &A, const Blk<KT, JT> &B, Blk<IT, JT> &C) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int jj = 0; jj < JT; ++jj)
#pragma unroll
            for (int k = 0; k < KT; ++k) C.t[i][jj] = mm4(A.t[k][i], B.t[k][jj], C.t[i][jj]);
}
// Loads are unconditional: addresses clamped into the array by integer minima, values multiplied by 0 / 1 masks (wide16.h).
// element (i, jj) = X[i + ld jj] of a rows x cols column-major matrix; `pad` on the diagonal outside it
template <int RT, int CT>
__device__ __forceinline__ void ld_blk(Blk<RT, CT> &o, const double *Xg, const int ld, const int rows, const int cols, const double pad, const int g, const int j) {
    const gbld *const X = (const gbld *)Xg;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) {
            const int jj = 16 * b + j, jc = min(jj, cols - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + 4 * r + g;
                const bool in = i < rows && jj < cols;
                o.t[a][b][r] = fma(X[min(i, rows - 1) + ld * jc], in ? 1.0 : 0.0, (!in && i == jj) ? pad : 0.0);
            }
        }
}
// element (i, jj) = X[jj + ld i]: the transpose of a column-major matrix with `cols` rows and `rows` columns
template <int RT, int CT>
__device__ __forceinline__ void ld_blk_T(Blk<RT, CT> &o, const double *Xg, const int ld, const int rows, const int cols, const int g, const int j) {
    const gbld *const X = (const gbld *)Xg;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) {
            const int jj = 16 * b + j, jc = min(jj, cols - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + 4 * r + g;
                o.t[a][b][r] = X[jc + ld * min(i, rows - 1)] * ((i < rows && jj < cols) ? 1.0 : 0.0);
            }
        }
}
// a block from its register image (wide.h: WideProblemDev.t*): coalesced loads at immediate offsets, nothing to clamp or mask
template <int RT, int CT>
__device__ __forceinline__ void ld_img(Blk<RT, CT> &o, const double *img, const int l) {
    const gbld *const X = (const gbld *)img + l;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) o.t[a][b][r] = X[((a * CT + b) * 4 + r) * 64];
}
// a vector of length len into column 0
template <int RT>
__device__ __forceinline__ void ld_col0(Blk<RT, 1> &o, const double *vg, const int len, const int g, const int j) {
    const gbld *const v = (const gbld *)vg;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * a + 4 * r + g;
            o.t[a][0][r] = v[min(i, len - 1)] * ((j == 0 && i < len) ? 1.0 : 0.0);
        }
}

// One 2 x 2 block-pivot round of the symmetric sweep on a TT x TT block matrix (device_utils.h: elim_round, one tile).  Pivot rows k, k + 1
// (k = 2 KB) live in row tile ap = KB / 8, register kr, quad-rows kg, kg + 1 of that tile.
template <int KB, int TT>
__device__ __forceinline__ void elim32_round(Blk<TT, TT> &M, const ldsd *const mk, const double (&es)[4], const bool odd, int &pdmin, double &rprod) {
    constexpr int ap = KB / 8, kq = KB % 8, k = 2 * kq, kr = k >> 2, kg = k & 3;
    const double tm = mk[64 * W16_TM(kq)], wa = mk[64 * W16_WA(kq)], cm = mk[64 * W16_CM(kq)], crm = mk[64 * W16_CRM(kq)];
    const double p11 = readlane_f64(M.t[ap][ap][kr], kg * 16 + k);
    const double p12 = readlane_f64(M.t[ap][ap][kr], kg * 16 + k + 1);
    const double p22 = readlane_f64(M.t[ap][ap][kr], (kg + 1) * 16 + k + 1);
    const double det = fma(p11, p22, -(p12 * p12));
    const double idet = fast_rcp1(det);
    pdmin = min(pdmin, min(__double2hiint(p11), __double2hiint(det)));
    rprod *= det;
    const double pd = es[kg] * p22 + es[kg + 1] * p11;
    const double rowsel = es[kg] + es[kg + 1], rowz = 1.0 - rowsel;      // 1 / 0 on the pivot rows' lanes
    double t[TT], nu[TT];
#pragma unroll
    for (int b = 0; b < TT; ++b) {
        t[b] = (b == ap) ? fma(M.t[ap][b][kr], tm, wa) : M.t[ap][b][kr] * rowsel;      // pivot rows; -I in the pivot block
        const double other = row_partner<0>(t[b], odd);
        nu[b] = fma(p12, other, -(pd * t[b])) * idet;                                  // -(Bk t) on the pivot-row lanes
    }
#pragma unroll
    for (int a = 0; a < TT; ++a)
#pragma unroll
        for (int b = 0; b < TT; ++b) {
            if (b == ap) {                                   // pivot columns (and, in the pivot rows' register, the pivot rows) cleared
#pragma unroll
                for (int r = 0; r < 4; ++r) M.t[a][b][r] *= (a == ap && r == kr) ? crm : cm;
            } else if (a == ap) {
                M.t[a][b][kr] *= rowz;                       // pivot rows cleared
            }
        }
    // (the tiles the next round reads first are updated first)
#pragma unroll
    for (int a = 0; a < TT; ++a)
#pragma unroll
        for (int b = 0; b < TT; ++b) M.t[a][b] = MFMA(t[a], nu[b], M.t[a][b]);
}
// (rounds whose pivot rows lie beyond the matrix -- unit diagonal, nothing coupled -- are skipped: what they would leave there, -1 instead of
//  1 on the padded diagonal, only ever meets zero rows)
template <int KB, int TT>
__device__ __forceinline__ void elim32_rounds(Blk<TT, TT> &M, const ldsd *const mk, const double (&es)[4], const bool odd, int &pdmin, double &rprod, int &rexp,
                                              const int size) {
    if (2 * KB < size) elim32_round<KB, TT>(M, mk, es, odd, pdmin, rprod);
    if constexpr (KB % 8 == 7) {                             // (the determinant product is renormalised once per row tile)
        if (rprod * 0.0 == 0.0) { rexp += __builtin_amdgcn_frexp_exp(rprod); rprod = __builtin_amdgcn_frexp_mant(rprod); }
    }
    W32_FENCE();
    if constexpr (KB + 1 < 8 * TT) elim32_rounds<KB + 1, TT>(M, mk, es, odd, pdmin, rprod, rexp, size);
}

// approximate_model's cost gradients (:294-313) of a whole trajectory, SIXTEEN steps per product: with the steps as the columns of a tile,
// c_x = Q X + P'U + q_vec and c_u = R U + P X + r_vec are four block products per 16 steps instead of four one-column products (which use 1 / 16
// of a tile) per step of every sweep that reads the trajectory.  Time-invariant cost tables only (the caller checks).  gq [N][n], gr [N][m]:
// the gradients; gc [64]: the lanes' parts of sum_t c(x_t, u_t) - q0 (the sweeps add them to their own per-lane sums).
template <int NT, int MT>
__device__ __noinline__ void grad32(const WideProblemDev &pb_in, const double *const x_, const double *const u_, double *const gq_, double *const gr_,
                                    double *const gc_) {
    const WideProblemDev pb = pb_in;
    const gbld *const x = (const gbld *)x_, *const u = (const gbld *)u_;
    gbld *const gq = (gbld *)gq_, *const gr = (gbld *)gr_;
    const int n = pb.n, m = pb.m, N = pb.N;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    double acc = 0.0;
    for (int t0 = 0; t0 < N; t0 += 16) {
        const int t = t0 + j, tc = min(t, N - 1);
        const bool tin = t < N;
        Blk<NT, 1> Xb, qx, qvb;
        Blk<MT, 1> Ub, ru, px, rvb;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + 4 * r + g;
                const bool in = tin && i < n;
                Xb.t[a][0][r] = (x + (size_t)tc * n)[min(i, n - 1)] * (in ? 1.0 : 0.0);
                qvb.t[a][0][r] = ((const gbld *)pb.qv)[min(i, n - 1)] * (in ? 1.0 : 0.0);
            }
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * c + 4 * r + g;
                const bool in = tin && i < m;
                Ub.t[c][0][r] = (u + (size_t)tc * m)[min(i, m - 1)] * (in ? 1.0 : 0.0);
                rvb.t[c][0][r] = ((const gbld *)pb.rv)[min(i, m - 1)] * (in ? 1.0 : 0.0);
            }
        blk_zero(qx); blk_zero(ru); blk_zero(px);
        {
            Blk<NT, NT> Q;
            ld_img(Q, pb.tQ, l);
            pmm(Q, Xb, qx);
        }
        Blk<NT, 1> cq = qx;
        {
            Blk<MT, NT> Pm;
            ld_img(Pm, pb.tP, l);
            pmm(Pm, Ub, cq);
        }
        {
            Blk<MT, MT> R;
            ld_img(R, pb.tR, l);
            pmm(R, Ub, ru);                                     // (unit diagonal beyond m: times u = 0 there)
        }
        {
            Blk<NT, MT> PT;
            ld_img(PT, pb.tPT, l);
            pmm(PT, Xb, px);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + 4 * r + g;
                acc += Xb.t[a][0][r] * (0.5 * qx.t[a][0][r] + qvb.t[a][0][r]);
                if (tin && i < n) (gq + (size_t)t * n)[i] = cq.t[a][0][r] + qvb.t[a][0][r];
            }
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * c + 4 * r + g;
                acc += Ub.t[c][0][r] * (0.5 * ru.t[c][0][r] + px.t[c][0][r] + rvb.t[c][0][r]);
                if (tin && i < m) (gr + (size_t)t * m)[i] = (ru.t[c][0][r] + px.t[c][0][r]) + rvb.t[c][0][r];
            }
    }
    ((gbld *)gc_)[l] = acc;
}

// solve_approximate_dp (GAIN = false, :412-465) / one pass of solve_approximate_dp! (GAIN = true, :341-406) over the trajectory (x, u) of an
// LQ-family problem with n <= 16 NT, m <= 16 MT.  Returns 0, 2 (M not positive definite) or -1 (H not positive definite: the caller raises
// mu and restarts).  tab: the 0 / 1 tables of setup16 in LDS (the rounds' masks depend on the position inside a tile only).
template <bool GAIN, bool ZEROL, int NT, int MT>
__device__ __noinline__ int sweep32(const WideProblemDev &pb_in, const ldsd *const tab, const double *const x_, const double *const u_,
                                       const double theta, const double mu, double *const Lg_, double *const dlg_out, double &value,
                                       const double *const gq_, const double *const gr_, const double *const gc_) {
    const WideProblemDev pb = pb_in;
    const gbld *const x = (const gbld *)x_, *const u = (const gbld *)u_;
    gbld *const Lg = (gbld *)Lg_, *const dlg = (gbld *)dlg_out;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const bool odd = (g & 1) != 0;
    const double es[4] = {g == 0 ? 1.0 : 0.0, g == 1 ? 1.0 : 0.0, g == 2 ? 1.0 : 0.0, g == 3 ? 1.0 : 0.0};
    const ldsd *const mk = tab + l;
    const double k3 = 3.0 * pb.kappa;
    const double coef = (theta != 0.0) ? -1.0 / (2.0 * theta) : 0.0;
    const double nth = -theta;

    // terminal condition (:352-354 / :429-431)
    Blk<NT, NT> S;
    Blk<NT, 1> sv;
    double acc = 0.0;                                           // per-lane parts of the scalar s (stage costs, 0.5 dl'H dl + dl'g)
    ld_img(S, pb.tQf, l);
    {
        Blk<NT, 1> xv, qf, qx;
        ld_col0(xv, x_ + (size_t)N * n, n, g, j);
        ld_img(qf, pb.tqvf, l);
        blk_zero(qx);
        pmm(S, xv, qx);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sv.t[a][0][r] = qx.t[a][0][r] + qf.t[a][0][r]; acc += xv.t[a][0][r] * (0.5 * qx.t[a][0][r] + qf.t[a][0][r]); }
    }
    // wave-uniform part of s: the q0's of all steps, summed up front
    double usum;
    {
        double p0 = 0.0;
        for (int t = l; t < N; t += 64) p0 += ((const gbld *)pb.q0)[pb.cost_tv ? t : 0];
        usum = wsum(p0) + pb.q0f;
    }
    // logdet(W M) = logdet W(k) + logdet M is formed PER STEP (two numbers of magnitude n log(1 / w) that cancel to O(theta tr(W S))): summed
    // over the horizon first, the two would meet at fifty times the magnitude and lose its rounding -- which -1 / (2 theta) then amplifies
    double racc = 0.0, rprod = 1.0, lsum = 0.0;
    int rexp = 0;
    // (x_t, u_t) of the step after this one are fetched while this one runs (raw values: masked where consumed); a policy evaluation's L_t is
    // fetched inside its own step, two phases ahead of its use -- nothing dynamic is carried across the elimination
    double xr_n[NT][4], ur_n[MT][4];
    auto fetch = [&](const int t, const int g) {
        const gbld *const xt = x + (size_t)t * n, *const ut = u + (size_t)t * m;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) xr_n[a][r] = xt[min(16 * a + 4 * r + g, n - 1)];
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) ur_n[c][r] = ut[min(16 * c + 4 * r + g, m - 1)];
    };
    fetch(N - 1, g);
    const int g_ = g, j_ = j;
    for (int t = N - 1; t >= 0; --t) {
        // (opaque per-step copies of the lane indices: the 0 / 1 masks and clamped offsets of the step's dynamic loads and stores are formed
        //  where they are used instead of being kept, four per tile, as loop invariants)
        int g = g_, j = j_;
        asm volatile("" : "+v"(g), "+v"(j));
        const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
        const double *const tQk = pb.tQ + (size_t)kc * (NT * NT * WIDE_IMG_TILE), *const tPk = pb.tP + (size_t)kc * (MT * NT * WIDE_IMG_TILE),
                     *const tRk = pb.tR + (size_t)kc * (MT * MT * WIDE_IMG_TILE);
        // this step's x_t on every lane of its rows (f_x's diagonal)
        double xrow[NT][4];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) xrow[a][r] = xr_n[a][r] * ((16 * a + 4 * r + g < n) ? 1.0 : 0.0);
        // approximate_model at (x_t, u_t) (:294-313): q_vec = Q x + P'u + q_vec, r_vec = R u + P x + r_vec, c -- read back from grad32's pass over
        // the trajectory (time-invariant cost), or formed here as one-column products (time-varying tables)
        Blk<NT, 1> qvt;
        Blk<MT, 1> rvt;
        if (!pb.cost_tv) {
            const gbld *const gqt = (const gbld *)gq_ + (size_t)t * n, *const grt = (const gbld *)gr_ + (size_t)t * m;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int i = 16 * a + 4 * r + g; qvt.t[a][0][r] = gqt[min(i, n - 1)] * ((j == 0 && i < n) ? 1.0 : 0.0); }
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int i = 16 * c + 4 * r + g; rvt.t[c][0][r] = grt[min(i, m - 1)] * ((j == 0 && i < m) ? 1.0 : 0.0); }
        } else {
            Blk<NT, 1> xv;
            Blk<MT, 1> uv;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) xv.t[a][0][r] = (j == 0) ? xrow[a][r] : 0.0;
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) uv.t[c][0][r] = ur_n[c][r] * ((j == 0 && 16 * c + 4 * r + g < m) ? 1.0 : 0.0);
            {
                Blk<NT, NT> Q;
                Blk<NT, 1> qx, qvc;
                ld_img(Q, tQk, l);
                ld_img(qvc, pb.tqv + (size_t)kc * (NT * WIDE_IMG_TILE), l);
                blk_zero(qx);
                pmm(Q, xv, qx);
                qvt = qx;
                Blk<MT, NT> Pm;                                     // P, natural rows (m x n)
                ld_img(Pm, tPk, l);
                pmm(Pm, uv, qvt);                                   // + P'u
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qvt.t[a][0][r] += qvc.t[a][0][r]; acc += xv.t[a][0][r] * (0.5 * qx.t[a][0][r] + qvc.t[a][0][r]); }
            }
            W32_FENCE();
            {
                Blk<MT, MT> R;
                Blk<NT, MT> PT;                                     // P' (n x m)
                Blk<MT, 1> ru, px, rvc;
                ld_img(R, tRk, l);                                  // unit diagonal beyond m
                ld_img(PT, pb.tPT + (size_t)kc * (MT * NT * WIDE_IMG_TILE), l);
                ld_img(rvc, pb.trv + (size_t)kc * (MT * WIDE_IMG_TILE), l);
                blk_zero(ru); blk_zero(px);
                pmm(R, uv, ru);
                pmm(PT, xv, px);
#pragma unroll
                for (int c = 0; c < MT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rvt.t[c][0][r] = (ru.t[c][0][r] + px.t[c][0][r]) + rvc.t[c][0][r];
                        acc += uv.t[c][0][r] * (0.5 * ru.t[c][0][r] + px.t[c][0][r] + rvc.t[c][0][r]);
                    }
            }
        }
        W32_FENCE();
        // f_x = A + 3 kappa diag(x^2), f_u = B: fetched where they are used (twice: a block of four tiles is 32 registers across the elimination)
        auto load_fx = [&](Blk<NT, NT> &Ad) {
            ld_img(Ad, pb.tA, l);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ad.t[a][a][r] = fma((4 * r + g == j) ? 1.0 : 0.0, k3 * (xrow[a][r] * xrow[a][r]), Ad.t[a][a][r]);
        };
        // T = (D S)[A | B] and D s_vec, with D S = S + theta S M^-1 S formed once (two block products) instead of X = S [A | B], Y = theta M^-1 X,
        // T = X + S Y (three products over [A | B]'s NT + MT column tiles): 32 ... 64 MFMAs fewer per step
        Blk<NT, NT> T1;
        Blk<NT, MT> TB;
        Blk<NT, 1> Ta = sv;
        blk_zero(T1); blk_zero(TB);
        if (theta != 0.0) {
            Blk<NT, NT> DS = S;
            {
                Blk<NT, NT> M;
                ld_img(M, pb.tWinv + (size_t)kw * (NT * NT * WIDE_IMG_TILE), l);   // unit pivots beyond n: det 1, nothing coupled
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) M.t[a][b][r] = fma(nth, S.t[a][b][r], M.t[a][b][r]);       // M = Symmetric(inv(W) - theta S)   (:365)
                int pdmin = 1;
                elim32_rounds<0, NT>(M, mk, es, odd, pdmin, rprod, rexp, n);
                if (!(pdmin > 0) || !(rprod * 0.0 == 0.0)) return 2;               // @assert isposdef(M)  (:366 / :440)
                lsum += ((const gbld *)pb.ldW)[kw] + (log(rprod) + (double)rexp * 0.6931471805599453094);      // logdet(W M)   (:387)
                rprod = 1.0; rexp = 0;
                W32_FENCE();
                {
                    Blk<NT, NT> U;
                    blk_zero(U);
                    pmm(M, S, U);                                                  // -M^-1 S
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r) U.t[a][b][r] *= nth;        // theta M^-1 S
                    pmm(S, U, DS);                                                  // D S = S + theta S M^-1 S   (:367)
                }
                W32_FENCE();
                {
                    Blk<NT, 1> Y;
                    blk_zero(Y);
                    pmm(M, Ta, Y);
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { Y.t[a][0][r] *= nth; racc += sv.t[a][0][r] * Y.t[a][0][r]; }      // theta s_vec'M^-1 s_vec  (:387)
                    pmm(S, Y, Ta);                                                  // D s_vec = s_vec + theta S M^-1 s_vec
                }
            }
            W32_FENCE();
            {
                Blk<NT, NT> Ad;
                load_fx(Ad);
                pmm(DS, Ad, T1);                                // (D S) A   (D S symmetric)
            }
            {
                Blk<NT, MT> Z2;
                ld_img(Z2, pb.tB, l);
                pmm(DS, Z2, TB);
            }
        } else {
            // theta == 0: D = I; 0.5 tr(W S)  (:385).  The reference still asserts isposdef(inv(W) - 0 S): a non-finite S fails it.
            double nf = 0.0;
            {
                Blk<NT, NT> Wt;
                ld_img(Wt, pb.tW + (size_t)kw * (NT * NT * WIDE_IMG_TILE), l);
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { nf = fma(S.t[a][b][r], 0.0, nf); racc = fma(Wt.t[a][b][r], S.t[a][b][r], racc); }
            }
            if (__ballot(nf != nf) != 0ull) return 2;
            {
                Blk<NT, NT> Ad;
                load_fx(Ad);
                pmm(S, Ad, T1);
            }
            {
                Blk<NT, MT> Z2;
                ld_img(Z2, pb.tB, l);
                pmm(S, Z2, TB);
            }
        }
        W32_FENCE();
        // (S is dead from here: its registers take the new value function)
        // a policy evaluation's gains: fetched now, consumed two phases on
        double Lr[MT][NT][4];
        if (!GAIN && !ZEROL) {
            const gbld *const Lt = Lg + (size_t)t * nm;
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Lr[c][b][r] = Lt[min(16 * c + 4 * r + g, m - 1) + m * min(16 * b + j, n - 1)];
        }
        // F = [A | B]'T + the step's cost model
        Blk<NT, NT> F11;
        Blk<NT, 1> F1a = qvt;
        Blk<MT, NT> G;
        Blk<MT, MT> H;
        Blk<MT, 1> gv = rvt;
        {
            Blk<NT, NT> Ad;
            load_fx(Ad);
            ld_img(F11, tQk, l);
            pmm(Ad, T1, F11);                                   // Q + A'(D S)A   (:390)
            pmm(Ad, Ta, F1a);                                   // q_vec + A'D s_vec   (:389)
        }
        W32_FENCE();
        {
            Blk<NT, MT> Z2;
            ld_img(Z2, pb.tB, l);
            ld_img(G, tPk, l);
            pmm(Z2, T1, G);                                     // P + B'(D S)A   (:369)
            ld_img(H, tRk, l);
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) H.t[c][c][r] += (4 * r + g == j) ? mu : 0.0;
            pmm(Z2, TB, H);                                     // R + B'(D S)B + mu I   (:370)
            pmm(Z2, Ta, gv);                                    // r + B'D s_vec   (:368)
        }
        W32_FENCE();
        fetch(t > 0 ? t - 1 : 0, g);                            // the next step's (x, u): two phases ahead of their use
        Blk<MT, NT> L;
        Blk<MT, 1> dl;
        blk_zero(dl);
        if (GAIN) {
            Blk<MT, MT> Hi = H;
            int pdh = 1, hexp = 0;
            double hprod = 1.0;
            elim32_rounds<0, MT>(Hi, mk, es, odd, pdh, hprod, hexp, m);
            if (!(pdh > 0) || !(hprod * 0.0 == 0.0)) return -1;                 // isposdef(H) fails   (:372)
            blk_zero(L);
            pmm(Hi, G, L);                                      // [L | dl] = -H \ [G | g]   (:379-382)
            pmm(Hi, gv, dl);
#pragma unroll
            for (int c = 0; c < MT; ++c) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = 16 * c + 4 * r + g, si = 16 * b + j;
                        if (ci < m && si < n) (Lg + (size_t)t * nm)[ci + m * si] = L.t[c][b][r];
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 16 * c + 4 * r + g;
                    if (j == 0 && ci < m) (dlg + (size_t)t * m)[ci] = dl.t[c][0][r];
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        L.t[c][b][r] = ZEROL ? 0.0 : Lr[c][b][r] * ((16 * c + 4 * r + g < m && 16 * b + j < n) ? 1.0 : 0.0);
        }
        W32_FENCE();
        // S = Q + A'(D S)A + L'(H L + G) + G'L,  s_vec = q_vec + A'D s_vec + L'(H dl + g) + G'dl   (:389-391)
        {
            Blk<MT, NT> U = G;
            pmm(H, L, U);                                       // H L + G
            pmm(L, U, F11);
            pmm(G, L, F11);
        }
        {
            Blk<MT, 1> hd;
            blk_zero(hd);
            pmm(H, dl, hd);                                     // H dl
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc += dl.t[c][0][r] * (0.5 * hd.t[c][0][r] + gv.t[c][0][r]);       // 0.5 dl'H dl + dl'g   (:383)
                    hd.t[c][0][r] += gv.t[c][0][r];
                }
            pmm(L, hd, F1a);
            pmm(G, dl, F1a);
        }
        S = F11; sv = F1a;
        W32_FENCE();
    }
    if (!pb.cost_tv) acc += ((const gbld *)gc_)[l];            // the stage costs of the trajectory (grad32)
    double tot = acc + 0.5 * racc;
    tot = wsum(tot) + usum;
    if (theta != 0.0) tot += coef * lsum;                                                           // -(logdet W + logdet M) / (2 theta)
    value = tot;
    return 0;
}

// simulate_dynamics in the same block form (wide16.h: rollout16): x_t is a one-column block,
//   u_t = l_t + eps dl_t + L_t (x_t - xbar_t)    one-column product on L_t' (rows = states)                                      ileqg.jl:82
//   x_{t+1} = A x_t + B u_t + kappa x_t^3        one-column products on A' (they do not wait for u_t) and B'
// CLOSED = false: simulate_dynamics(problem, x_0, u_array) (:18-38): xbar_ = x_0, l_ = u_array, no gains.  The operands of step t + 1 are
// fetched at the top of step t.  Returns maximum(norm.(l .- u_new)) (:539).
template <bool CLOSED, int NT, int MT>
__device__ __noinline__ double rollout32(const WideProblemDev &pb_in, const double *const xbar_, const double *const l_, const double *const dl_,
                                            const double *const L_, const double eps, double *const xo_, double *const uo_) {
    const WideProblemDev pb = pb_in;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const gbld *const xbar = (const gbld *)xbar_, *const lg = (const gbld *)l_, *const dlg = (const gbld *)dl_, *const Lg = (const gbld *)L_;
    gbld *const xo = (gbld *)xo_, *const uo = (gbld *)uo_;
    Blk<NT, NT> AT;                                             // A'
    Blk<MT, NT> BT;                                             // B', natural rows (m x n)
    ld_img(AT, pb.tAT, l);
    ld_img(BT, pb.tBT, l);
    Blk<NT, 1> xv;
    ld_col0(xv, xbar_, n, g, j);
    if (j == 0) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (16 * a + 4 * r + g < n) xo[16 * a + 4 * r + g] = xv.t[a][0][r];
    }
    double dmax = -INFINITY;
    bool dnan = false;
    double LTr[NT][MT][4], xbr[NT][4], lr[MT][4], dlr[MT][4];
    auto issue = [&](const int tq) {
        const int t = (tq < N) ? tq : N - 1;
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = min(16 * c + 4 * r + g, m - 1);
                lr[c][r] = (lg + (size_t)t * m)[ci];
                dlr[c][r] = (CLOSED && dlg) ? (dlg + (size_t)t * m)[ci] : 0.0;
            }
        if (CLOSED) {
            const gbld *const Lt = Lg + (size_t)t * nm, *const xt = xbar + (size_t)t * n;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int si = min(16 * a + 4 * r + g, n - 1);
                    xbr[a][r] = xt[si];
#pragma unroll
                    for (int c = 0; c < MT; ++c) LTr[a][c][r] = Lt[min(16 * c + j, m - 1) + m * si];
                }
        }
    };
    issue(0);
    for (int t = 0; t < N; ++t) {
        Blk<NT, MT> LT;                                         // L_t' (n x m)
        Blk<NT, 1> dxv;
        Blk<MT, 1> lt, un;
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double mc = (j == 0 && 16 * c + 4 * r + g < m) ? 1.0 : 0.0;
                lt.t[c][0][r] = lr[c][r] * mc;
                un.t[c][0][r] = lt.t[c][0][r] + eps * (dlr[c][r] * mc);
            }
        if (CLOSED) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool rin = 16 * a + 4 * r + g < n;
                    dxv.t[a][0][r] = xv.t[a][0][r] - xbr[a][r] * ((j == 0 && rin) ? 1.0 : 0.0);
#pragma unroll
                    for (int c = 0; c < MT; ++c) LT.t[a][c][r] = LTr[a][c][r] * ((rin && 16 * c + j < m) ? 1.0 : 0.0);
                }
        }
        issue(t + 1);
        Blk<NT, 1> xa;
        blk_zero(xa);
        pmm(AT, xv, xa);                                        // A x_t: does not wait for the feedback control
        if (CLOSED) {
            pmm(LT, dxv, un);                                   // + L_t (x_t - xbar_t)
            double dsq = 0.0;
#pragma unroll
            for (int c = 0; c < MT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double df = lt.t[c][0][r] - un.t[c][0][r]; dsq = fma(df, df, dsq); }
            const double dn2 = wsum(dsq);
            dnan |= (dn2 != dn2);
            dmax = (dn2 > dmax) ? dn2 : dmax;                   // (sqrt is monotone: rooted once after the loop)
        }
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = 16 * c + 4 * r + g;
                if (j == 0 && ci < m) (uo + (size_t)t * m)[ci] = un.t[c][0][r];
            }
        pmm(BT, un, xa);                                        // + B u_t
        gbld *const xn = xo + (size_t)(t + 1) * n;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xc = xv.t[a][0][r];
                xv.t[a][0][r] = (j == 0) ? fma(pb.kappa, xc * xc * xc, xa.t[a][0][r]) : 0.0;
                if (j == 0 && 16 * a + 4 * r + g < n) xn[16 * a + 4 * r + g] = xv.t[a][0][r];
            }
    }
    return dnan ? NAN : sqrt(dmax);
}
