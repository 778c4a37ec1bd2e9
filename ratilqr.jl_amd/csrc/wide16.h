// wide16.h -- the backward sweep of wide.hip for n <= 16, m <= 4 (the first sizes beyond the 12 + 4 tile), entirely in registers on the
// matrix pipe: the formulation of kernels.hip's sweep_body one tile size up.  Included by wide.hip (inside its anonymous namespace).
//
// Everything is a 16 x 16 tile in the accumulator layout of v_mfma_f64_16x16x4_f64 (d4: register r of lane (g, j) = element (4 r + g, j)),
// and the one product primitive is P(a, b) = a'b = sum over the four K-slices s of MFMA(a[s], b[s]) -- register s of a matrix is the A operand
// of slice s of a'(.) and the B operand of slice s of (.)'b, so a step chains through registers without moving data between lanes.
// The state block and the control / affine block of a quantity are two tiles: "1" = columns of the n states, "2" = columns 0..3 of the m
// controls and column 4 of the affine part (s_vec, q_vec + ..., g ride along in column 4 of products that are needed for B anyway).
//   M = inv(W) - theta S                  -M^-1 by the symmetric sweep with 2 x 2 block pivots (8 rounds, one MFMA each; isposdef(M) <=> all
//                                         leading minors > 0; logdet from the running product of the block determinants)      ileqg.jl:365-366
//   X1 = S A, X2 = S [B | 0] + [0 | s_vec]      Y = theta M^-1 X      T = X + S Y = (D S)[A | B | S^-1 s_vec]: column 4 of T2 = D s_vec  :367
//   F11 = Q + A'T1   F12 = A'T2 = [(G - P)' | q_vec + A'D s_vec]   F22 = [R + mu I + B'T2 | r + B'D s_vec] = [H | g]               :368-371
//   H [L | dl] = -[G | g] by LDL' in every lane for its own column (pivots > 0 <=> isposdef(H))                                :372-382
//   S' = F11 + L'(H L + G) + G'L,  s_vec' = F12 + L'(H dl + g) + G'dl  (two rank-4 MFMAs each);  scalars accumulate per lane    :383-391
// The step's cost gradients (approximate_model, :294-313) are five + five MFMAs on [x_t; u_t] held in column 4.
// 58 MFMAs + ~0.5 k vector instructions per step against ~5 k instructions and a dozen LDS round trips in the general sweep().
#pragma once

// Per-lane 0 / 1 tables of the elimination rounds (tm, wa, cm, crm of ElimMasks, device_utils.h, for eight rounds), the one-hot diagonal
// selectors and W (read by theta == 0 steps only) live in LDS, 64 consecutive doubles per entry: 40 entries would be 80 registers, and the
// step already keeps ~150 values live.  W16_LDS doubles behind the general kernel's own area (wide_lds_bytes).
#define W16_TM(kb) (kb)
#define W16_WA(kb) (8 + (kb))
#define W16_CM(kb) (16 + (kb))
#define W16_CRM(kb) (24 + (kb))
#define W16_DG(r) (32 + (r))
#define W16_WP(r) (36 + (r))
#define W16_ENTRIES 40
#define W16_LDS (W16_ENTRIES * 64)

__device__ __forceinline__ d4 mm4(const d4 &a, const d4 &b, d4 acc) {       // acc += a'b over the four K-slices
    acc = MFMA(a[0], b[0], acc);
    acc = MFMA(a[1], b[1], acc);
    acc = MFMA(a[2], b[2], acc);
    acc = MFMA(a[3], b[3], acc);
    return acc;
}

struct RoundMasks { double tm, wa, cm, crm; };
__device__ __forceinline__ RoundMasks round_masks(const ldsd *const mk, const int kb) {
    RoundMasks r;
    r.tm = mk[64 * W16_TM(kb)]; r.wa = mk[64 * W16_WA(kb)]; r.cm = mk[64 * W16_CM(kb)]; r.crm = mk[64 * W16_CRM(kb)];
    return r;
}
template <int KB>
__device__ __forceinline__ void elim16_round(d4 &m, const RoundMasks &rm, const double (&es)[4], const bool odd, int &pdmin, double &rprod) {
    constexpr int k = 2 * KB, kr = k >> 2, kg = k & 3;          // rows k, k+1 live in register kr, quad-rows kg, kg+1
    const double p11 = readlane_f64(m[kr], kg * 16 + k);
    const double p12 = readlane_f64(m[kr], kg * 16 + k + 1);
    const double p22 = readlane_f64(m[kr], (kg + 1) * 16 + k + 1);
    const double cm = rm.cm, crm = rm.crm;
    const double t = fma(m[kr], rm.tm, rm.wa);          // pivot rows, -I in the pivot block, zero elsewhere
    const double other = row_partner<0>(t, odd);
    const double det = fma(p11, p22, -(p12 * p12));
    const double idet = fast_rcp1(det);
    pdmin = min(pdmin, min(__double2hiint(p11), __double2hiint(det)));
    rprod *= det;
    const double pd = es[kg] * p22 + es[kg + 1] * p11;
    const double nu = fma(p12, other, -(pd * t)) * idet;
    m[0] *= (kr == 0 ? crm : cm);
    m[1] *= (kr == 1 ? crm : cm);
    m[2] *= (kr == 2 ? crm : cm);
    m[3] *= (kr == 3 ? crm : cm);
    m = MFMA(t, nu, m);
}

// the tables of one (problem, lane): once per solve
__device__ __forceinline__ void setup16(const WideProblemDev &pb, ldsd *const tab) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15, n = pb.n;
    ldsd *const mk = tab + l;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const int k = 2 * kb, kg = k & 3;
        const bool colk = (j == k) || (j == k + 1), rowk = (g == kg) || (g == kg + 1);
        mk[64 * W16_TM(kb)] = (rowk && !colk) ? 1.0 : 0.0;
        mk[64 * W16_WA(kb)] = (rowk && j == k + (g - kg)) ? -1.0 : 0.0;
        mk[64 * W16_CM(kb)] = colk ? 0.0 : 1.0;
        mk[64 * W16_CRM(kb)] = (colk || rowk) ? 0.0 : 1.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        mk[64 * W16_DG(r)] = (i == j && i < n) ? 1.0 : 0.0;
        const bool in = i < n && j < n;
        mk[64 * W16_WP(r)] = in ? pb.W[in ? i + n * j : 0] : 0.0;           // (time-varying W: reloaded by the steps that need it)
    }
    WAVE_SYNC();
}

// solve_approximate_dp (gain = false, :412-465) / one pass of solve_approximate_dp! (gain = true, :341-406) over the trajectory (x, u) of an
// LQ-family problem with n <= 16, m <= 4.  Returns 0, 2 (M not positive definite) or -1 (H not positive definite: the caller raises mu and
// restarts).  ex: 192 doubles of the workgroup's LDS (the [H | g] and G rows of a step, for the per-lane gain solve).
// (GAIN / ZEROL are template parameters and the function is inlined at its three call sites: the kernel's pointers keep their address space
//  -- through a real call every one of them is generic, each load a FLAT instruction that holds vmcnt AND lgkmcnt -- and the prefetch of the
//  next step's operands sits in straight-line code.)
typedef __attribute__((address_space(1))) double gbld;
template <bool GAIN, bool ZEROL>
__device__ __forceinline__ int sweep16(const WideProblemDev &pb_in, ldsd *const ex, const ldsd *const tab, const double *const x_, const double *const u_,
                                       const double theta, const double mu, double *const Lg_, double *const dlg_out, double &value) {
    const WideProblemDev pb = pb_in;
    constexpr bool gain = GAIN, zeroL = ZEROL;
    const gbld *const x = (const gbld *)x_, *const u = (const gbld *)u_;
    gbld *const Lg = (gbld *)Lg_, *const dlg = (gbld *)dlg_out;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m, n2 = n * n, mm = m * m;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const bool odd = (g & 1) != 0;
    const double es[4] = {g == 0 ? 1.0 : 0.0, g == 1 ? 1.0 : 0.0, g == 2 ? 1.0 : 0.0, g == 3 ? 1.0 : 0.0};
    const double c4 = (j == 4) ? 1.0 : 0.0;                     // the affine column of the "2" tiles
    const ldsd *const mk = tab + l;
    // Every load is unconditional: the address is clamped into the array by integer minima and the value multiplied by a 0 / 1 mask
    // (a select on a loaded value compiles to a divergent branch around the load, each with its own wait).
    bool in_nn[4], in_nm[4];
    int ri[4], rc[4];
    const int jn = min(j, n - 1), jm = min(j, m - 1), gm = min(g, m - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) { ri[r] = 4 * r + g; rc[r] = min(ri[r], n - 1); in_nn[r] = ri[r] < n && j < n; in_nm[r] = ri[r] < n && j < m; }
    auto ld_nn = [&](const double *Xg, const double pad) {      // an n x n column-major matrix; `pad` on the diagonal beyond n
        const gbld *const X = (const gbld *)Xg;
        d4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fma(X[rc[r] + n * jn], in_nn[r] ? 1.0 : 0.0, (!in_nn[r] && ri[r] == j) ? pad : 0.0);
        return o;
    };
    auto ld_vec4 = [&](const double *vg) {                      // an n-vector into column 4
        const gbld *const v = (const gbld *)vg;
        d4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = v[rc[r]] * ((j == 4 && ri[r] < n) ? 1.0 : 0.0);
        return o;
    };
    const bool in_mn = g < m && j < n, in_mm = g < m && j < m;
    const d4 A = ld_nn(pb.A, 0.0);
    d4 Z2;                                                      // [B | 0]
#pragma unroll
    for (int r = 0; r < 4; ++r) Z2[r] = ((const gbld *)pb.B)[rc[r] + n * jm] * (in_nm[r] ? 1.0 : 0.0);
    const double k3 = 3.0 * pb.kappa;
    d4 Winv, Q, PT, qvc;
    double Pn, Rn, rvc;
    auto load_cost = [&](const int kc) {
        Q = ld_nn(pb.Q + (size_t)kc * n2, 0.0);
        const gbld *const Pk = (const gbld *)pb.P + (size_t)kc * nm, *const Rk = (const gbld *)pb.R + (size_t)kc * mm;
        Pn = Pk[gm + m * jn] * (in_mn ? 1.0 : 0.0);                                                   // P, natural rows (m x n)
#pragma unroll
        for (int r = 0; r < 4; ++r) PT[r] = Pk[jm + m * rc[r]] * (in_nm[r] ? 1.0 : 0.0);              // P' (n x m)
        Rn = fma(Rk[gm + m * jm], in_mm ? 1.0 : 0.0, (!in_mm && g == j && g >= m) ? 1.0 : 0.0);       // R; unit diagonal beyond m
        qvc = ld_vec4(pb.qv + (size_t)kc * n);
        rvc = ((const gbld *)pb.rv)[(size_t)kc * m + gm] * ((j == 4 && g < m) ? 1.0 : 0.0);
    };
    auto load_noise = [&](const int kw) {
        Winv = ld_nn(pb.Winv + (size_t)kw * n2, 1.0);          // unit pivots beyond n: det 1, nothing coupled
    };
    if (!pb.cost_tv) load_cost(0);
    if (!pb.W_tv) load_noise(0);
    int hoff[4], gtoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gtoff[r] = (j < 4) ? 64 + j * 16 + 4 * r + g : 128 + l;   // (the lanes of columns >= 4 write to slots nobody reads)
#pragma unroll
    for (int c = 0; c < 4; ++c) hoff[c] = (g <= c) ? g * 16 + c : c * 16 + g;      // Symmetric(H): the upper triangle rules (:371)
    const double mH = (j == g) ? mu : 0.0;
    const double coef = (theta != 0.0) ? -1.0 / (2.0 * theta) : 0.0;
    const double nth = -theta;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};

    // terminal condition (:352-354 / :429-431)
    d4 S = ld_nn(pb.Qf, 0.0), sv2;
    double acc = 0.0;                                           // per-lane parts of the scalar s (stage costs, 0.5 dl'H dl + dl'g, risk terms)
    {
        d4 xv;
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = (x + (size_t)N * n)[rc[r]] * ((j == 4 && ri[r] < n) ? 1.0 : 0.0);
        const d4 qf = ld_vec4(pb.qvf);
        const d4 qx = mm4(S, xv, zero4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv2[r] = qx[r] + qf[r]; acc += xv[r] * (0.5 * qx[r] + qf[r]); }
    }
    // wave-uniform parts of s: the q0's and (theta != 0) the logdet W(k)'s of all steps, summed up front
    double usum, ldw;
    {
        double p0 = 0.0, p1 = 0.0;
        for (int t = l; t < N; t += 64) { p0 += ((const gbld *)pb.q0)[pb.cost_tv ? t : 0]; p1 += ((const gbld *)pb.ldW)[pb.W_tv ? t : 0]; }
        usum = wsum(p0) + pb.q0f;
        ldw = wsum(p1);
    }
    double racc = 0.0, rprod = 1.0;
    int rexp = 0;
    // (x_t, u_t) and, for a policy evaluation, L_t, dl_t of the step after this one are fetched while this one runs
    double xb_n[4], ug_n, Lc_n[4];
    // (raw values: the 0 / 1 masks are applied at the top of the step that consumes them, so nothing waits for a load in the step that issues it;
    //  a wave-uniform base per step and 32-bit lane offsets: global_load with a scalar base)
    const double mjn = (j < n) ? 1.0 : 0.0, mgm = (g < m) ? 1.0 : 0.0;
    double mrn[4];
    int offL[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrn[r] = (ri[r] < n) ? 1.0 : 0.0; offL[r] = min(r, m - 1) + m * jn; }
    auto fetch = [&](const int t) {
        const gbld *const xt = x + (size_t)t * n, *const ut = u + (size_t)t * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) xb_n[r] = xt[rc[r]];
        ug_n = ut[gm];
        if (!gain && !zeroL) {
            const gbld *const Lt = Lg + (size_t)t * nm;
#pragma unroll
            for (int c = 0; c < 4; ++c) Lc_n[c] = Lt[offL[c]];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) Lc_n[c] = 0.0;
        }
    };
    fetch(N - 1);
    for (int t = N - 1; t >= 0; --t) {
        double xb[4], Lc[4];
        const double ug = ug_n * mgm;
#pragma unroll
        for (int r = 0; r < 4; ++r) { xb[r] = xb_n[r] * mrn[r]; Lc[r] = Lc_n[r] * ((r < m) ? mjn : 0.0); }
        fetch(t > 0 ? t - 1 : 0);
        const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
        if (pb.cost_tv) load_cost(kc);
        if (pb.W_tv) load_noise(kw);
        // approximate_model at (x_t, u_t) (:294-313): f_x = A + 3 kappa diag(x^2); q_vec = Q x + P'u + q_vec, r_vec = R u + P x + r_vec, c
        d4 Ad, xv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { Ad[r] = fma(mk[64 * W16_DG(r)], k3 * (xb[r] * xb[r]), A[r]); xv[r] = c4 * xb[r]; }
        const double uv = c4 * ug;
        const d4 qx = mm4(Q, xv, zero4);
        d4 qvt = MFMA(Pn, uv, qx);
        const double ru = MFMA(Rn, uv, zero4)[0];
        const double px = mm4(PT, xv, zero4)[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) { qvt[r] += qvc[r]; acc += xv[r] * (0.5 * qx[r] + qvc[r]); }
        const double rvt = c4 * (ru + px) + rvc;              // (column 4, rows < m; the padded rows of R contribute u = 0)
        acc += uv * (0.5 * ru + px + rvc);
        // T = (D S)[A | B | S^-1 s_vec].  X = S [A | B] + [0 | s_vec] does not depend on the inverse: its eight MFMAs are issued one behind each
        // elimination round's own -- the round's result needs ~20 wait states before the next round may read it anyway, so they are free there
        d4 X1 = zero4, X2 = sv2;
        d4 T1, T2;
        if (theta != 0.0) {
            d4 M;
#pragma unroll
            for (int r = 0; r < 4; ++r) M[r] = fma(nth, S[r], Winv[r]);                 // M = Symmetric(inv(W) - theta S)   (:365)
            int pdmin = 1;
            // (the scheduler otherwise clumps the X products behind two of the rounds; with the rounds fenced, each round's 0 / 1 tables are
            //  fetched from LDS at the top of the round before it)
            RoundMasks ra = round_masks(mk, 0), rb;
            __builtin_amdgcn_sched_barrier(0);
            rb = round_masks(mk, 1); elim16_round<0>(M, ra, es, odd, pdmin, rprod); X1 = MFMA(S[0], Ad[0], X1); __builtin_amdgcn_sched_barrier(0);
            ra = round_masks(mk, 2); elim16_round<1>(M, rb, es, odd, pdmin, rprod); X1 = MFMA(S[1], Ad[1], X1); __builtin_amdgcn_sched_barrier(0);
            rb = round_masks(mk, 3); elim16_round<2>(M, ra, es, odd, pdmin, rprod); X1 = MFMA(S[2], Ad[2], X1); __builtin_amdgcn_sched_barrier(0);
            ra = round_masks(mk, 4); elim16_round<3>(M, rb, es, odd, pdmin, rprod); X1 = MFMA(S[3], Ad[3], X1); __builtin_amdgcn_sched_barrier(0);
            rb = round_masks(mk, 5); elim16_round<4>(M, ra, es, odd, pdmin, rprod); X2 = MFMA(S[0], Z2[0], X2); __builtin_amdgcn_sched_barrier(0);
            ra = round_masks(mk, 6); elim16_round<5>(M, rb, es, odd, pdmin, rprod); X2 = MFMA(S[1], Z2[1], X2); __builtin_amdgcn_sched_barrier(0);
            rb = round_masks(mk, 7); elim16_round<6>(M, ra, es, odd, pdmin, rprod); X2 = MFMA(S[2], Z2[2], X2); __builtin_amdgcn_sched_barrier(0);
            elim16_round<7>(M, rb, es, odd, pdmin, rprod); X2 = MFMA(S[3], Z2[3], X2); __builtin_amdgcn_sched_barrier(0);
            if (!(pdmin > 0) || !(rprod * 0.0 == 0.0)) return 2;                        // @assert isposdef(M)  (:366 / :440)
            rexp += __builtin_amdgcn_frexp_exp(rprod);
            rprod = __builtin_amdgcn_frexp_mant(rprod);
            d4 Y1 = mm4(M, X1, zero4), Y2 = mm4(M, X2, zero4);                          // -M^-1 X
#pragma unroll
            for (int r = 0; r < 4; ++r) { Y1[r] *= nth; Y2[r] *= nth; racc += sv2[r] * Y2[r]; }    // theta M^-1 X;  theta s_vec'M^-1 s_vec  (:387)
            T1 = mm4(S, Y1, X1);
            T2 = mm4(S, Y2, X2);
        } else {
            // theta == 0: D = I; 0.5 tr(W S)  (:385).  The reference still asserts isposdef(inv(W) - 0 S): a non-finite S fails it.
            const double nf = fma(S[3], 0.0, fma(S[2], 0.0, fma(S[1], 0.0, S[0] * 0.0)));
            if (__ballot(nf != nf) != 0ull) return 2;
            X1 = mm4(S, Ad, X1);
            X2 = mm4(S, Z2, X2);
            if (pb.W_tv) {
#pragma unroll
                for (int r = 0; r < 4; ++r) racc = fma(((const gbld *)pb.W)[(size_t)kw * n2 + rc[r] + n * jn] * (in_nn[r] ? 1.0 : 0.0), S[r], racc);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) racc = fma(mk[64 * W16_WP(r)], S[r], racc);
            }
            T1 = X1; T2 = X2;
        }
        // F = [A | B]'T + the step's cost model
        const d4 F11 = mm4(Ad, T1, Q);                                                  // Q + A'(D S)A   (:390)
        d4 F12 = mm4(Ad, T2, qvt);                                                      // column 4: q_vec + A'D s_vec   (:389)
        const d4 C22 = {Rn + mH + rvt, 0.0, 0.0, 0.0};
        const double hq = mm4(Z2, T2, C22)[0];                                          // [H | g] = [R + B'(D S)B + mu I | r + B'D s_vec]  (:368, :370)
        // G = P + B'(D S)A (:369) is the transpose of columns 0..3 of F12 = A'(D S)[B | .] (D S is symmetric): it goes through the exchange
        // the gain solve needs anyway -- written as (state, control), read back in natural rows -- instead of four MFMAs of its own
        ex[g * 16 + j] = hq;
#pragma unroll
        for (int r = 0; r < 4; ++r) ex[gtoff[r]] = F12[r] + PT[r];
        WAVE_SYNC();
        const double gq = ex[64 + g * 16 + j];
        const double hg0 = ex[hoff[0]], hg1 = ex[hoff[1]], hg2 = ex[hoff[2]], hg3 = ex[hoff[3]];       // row g of Symmetric(H)
        const double gvg = ex[g * 16 + 4];                                              // g_g
        double x0, x1, x2, x3, d0v, d1v, d2v, d3v;                                      // column j of L, and dl
        if (gain) {
            const double h00 = ex[0], h01 = ex[1], h02 = ex[2], h03 = ex[3], h11 = ex[17], h12 = ex[18], h13 = ex[19], h22 = ex[34], h23 = ex[35], h33 = ex[51];
            const double g0 = ex[64 + j], g1 = ex[80 + j], g2 = ex[96 + j], g3 = ex[112 + j];
            const double v0 = ex[4], v1 = ex[20], v2 = ex[36], v3 = ex[52];
            // LDL' of H; all pivots > 0 <=> isposdef(H)   (:372)
            const double d0 = h00, i0 = fast_rcp(d0);
            const double l10 = h01 * i0, l20 = h02 * i0, l30 = h03 * i0;
            const double d1 = h11 - l10 * h01, i1 = fast_rcp(d1);
            const double l21 = (h12 - l20 * h01) * i1, l31 = (h13 - l30 * h01) * i1;
            const double d2 = h22 - l20 * h02 - l21 * (l21 * d1), i2 = fast_rcp(d2);
            const double l32 = (h23 - l30 * h02 - l31 * (l21 * d1)) * i2;
            const double d3 = h33 - l30 * h03 - l31 * (l31 * d1) - l32 * (l32 * d2), i3 = fast_rcp(d3);
            if (!(d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) return -1;
            auto solve = [&](const double r0, const double r1, const double r2, const double r3, double &s0, double &s1, double &s2, double &s3) {
                const double y0 = -r0;                                                  // X = -H \ [G | g]   (:379-382)
                const double y1 = -r1 - l10 * y0;
                const double y2 = -r2 - l20 * y0 - l21 * y1;
                const double y3 = -r3 - l30 * y0 - l31 * y1 - l32 * y2;
                s3 = y3 * i3;
                s2 = y2 * i2 - l32 * s3;
                s1 = y1 * i1 - l21 * s2 - l31 * s3;
                s0 = y0 * i0 - l10 * s1 - l20 * s2 - l30 * s3;
            };
            solve(g0, g1, g2, g3, x0, x1, x2, x3);
            solve(v0, v1, v2, v3, d0v, d1v, d2v, d3v);
        } else {
            x0 = Lc[0]; x1 = Lc[1]; x2 = Lc[2]; x3 = Lc[3];
            d0v = d1v = d2v = d3v = 0.0;                   // (the solver evaluates u = l + L (x - xbar): no affine part, ileqg.jl:520-528)
        }
        const double la1 = ((x0 * es[0] + x1 * es[1]) + x2 * es[2]) + x3 * es[3];       // natural rows of L
        const double dlg_ = ((d0v * es[0] + d1v * es[1]) + d2v * es[2]) + d3v * es[3];   // dl_g
        const double la2 = c4 * dlg_;
        const double ua1 = hg0 * x0 + hg1 * x1 + hg2 * x2 + hg3 * x3 + gq;              // H L + G
        const double hd = hg0 * d0v + hg1 * d1v + hg2 * d2v + hg3 * d3v;                // H dl
        const double ua2 = c4 * (hd + gvg);
        acc += la2 * (0.5 * hd + gvg);                                                  // 0.5 dl'H dl + dl'g   (:383)
        if (gain) {
            if (in_mn) (Lg + (size_t)t * nm)[g + m * j] = la1;
            if (j == 4 && g < m) (dlg + (size_t)t * m)[g] = dlg_;
        }
        // S = Q + A'(D S)A + L'(H L + G) + G'L,  s_vec = q_vec + A'D s_vec + L'(H dl + g) + G'dl   (:389-391)
        d4 Sn = MFMA(la1, ua1, F11);
        Sn = MFMA(gq, la1, Sn);
#pragma unroll
        for (int r = 0; r < 4; ++r) F12[r] *= c4;
        d4 sn = MFMA(la1, ua2, F12);
        sn = MFMA(gq, la2, sn);
        S = Sn; sv2 = sn;
        WAVE_SYNC();                                                                    // the exchange area is rewritten next step
    }
    double tot = acc + 0.5 * racc;
    tot = wsum(tot) + usum;
    if (theta != 0.0) tot += coef * (ldw + log(rprod) + (double)rexp * 0.6931471805599453094);      // -(logdet W + logdet M) / (2 theta)
    value = tot;
    return 0;
}

// simulate_dynamics for n <= 16, m <= 4 in the same register form: x_t lives in column 4 of a tile (lane (g, 4), register r: x[4 r + g]), so
//   u_t = l_t + eps dl_t + L_t (x_t - xbar_t)    4 MFMAs on L_t' (rows = states), result in rows 0..3 of column 4                      ileqg.jl:82
//   x_{t+1} = A x_t + B u_t + kappa x_t^3        4 MFMAs on A' (they do not wait for u_t) + 1 on B' -- the result IS the next column 4
// closes the recursion without a cross-lane move (~0.5 k cycles per step against ~5 k for the LDS loops of rollout_closed, whose L_t loads
// were not prefetched).  CLOSED = false: simulate_dynamics(problem, x_0, u_array) (:18-38): xbar_ = x_0, l_ = u_array, no gains.
// Operands of step t + 3 are fetched at the top of step t into four rotating register sets.  Returns maximum(norm.(l .- u_new)) (:539).
template <bool CLOSED>
__device__ __forceinline__ double rollout16(const WideProblemDev &pb_in, const double *const xbar_, const double *const l_, const double *const dl_,
                                            const double *const L_, const double eps, double *const xo_, double *const uo_) {
    const WideProblemDev pb = pb_in;
    const int n = pb.n, m = pb.m, N = pb.N, nm = n * m;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const gbld *const xbar = (const gbld *)xbar_, *const lg = (const gbld *)l_, *const dlg = (const gbld *)dl_, *const Lg = (const gbld *)L_;
    gbld *const xo = (gbld *)xo_, *const uo = (gbld *)uo_;
    const double c4 = (j == 4) ? 1.0 : 0.0, mgm = (j == 4 && g < m) ? 1.0 : 0.0;
    const int jn = min(j, n - 1), jm = min(j, m - 1), gm = min(g, m - 1);
    int rc[4];
    double mrn[4], mLT[4];
    d4 AT;                                                      // A' : lane (g, j), register r = A[j][4 r + g]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        rc[r] = min(i, n - 1);
        mrn[r] = (j == 4 && i < n) ? 1.0 : 0.0;
        mLT[r] = (i < n && j < m) ? 1.0 : 0.0;
        AT[r] = ((const gbld *)pb.A)[jn + n * rc[r]] * ((i < n && j < n) ? 1.0 : 0.0);
    }
    const double BT = ((const gbld *)pb.B)[jn + n * gm] * ((g < m && j < n) ? 1.0 : 0.0);      // B', natural rows: lane (g, j) = B[j][g]
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    struct In { double LT[4], xb[4], l, dl; };
    In buf[4];
    auto issue = [&](In &in, const int tq) {
        const int t = (tq < N) ? tq : N - 1;
        in.l = (lg + (size_t)t * m)[gm];
        in.dl = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) in.LT[r] = in.xb[r] = 0.0;
        if (CLOSED) {
            if (dlg) in.dl = (dlg + (size_t)t * m)[gm];
            const gbld *const Lt = Lg + (size_t)t * nm, *const xt = xbar + (size_t)t * n;
#pragma unroll
            for (int r = 0; r < 4; ++r) { in.LT[r] = Lt[jm + m * rc[r]]; in.xb[r] = xt[rc[r]]; }
        }
    };
    d4 xv;                                                      // x_t
#pragma unroll
    for (int r = 0; r < 4; ++r) xv[r] = xbar[rc[r]] * mrn[r];
    if (j == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (4 * r + g < n) xo[4 * r + g] = xv[r];
    }
    double dmax = -INFINITY;
    bool dnan = false;
    issue(buf[0], 0); issue(buf[1], 1); issue(buf[2], 2);
    auto step = [&](const int t, const In &cur) {
        d4 xa = mm4(AT, xv, zero4);                             // A x_t: does not wait for the feedback control
        double un = cur.l * mgm;
        if (CLOSED) {
            d4 dxv, LT;
#pragma unroll
            for (int r = 0; r < 4; ++r) { dxv[r] = xv[r] - cur.xb[r] * mrn[r]; LT[r] = cur.LT[r] * mLT[r]; }
            const double fb = mm4(LT, dxv, zero4)[0];           // L_t (x_t - xbar_t), rows 0..3 of column 4
            const double lt = cur.l * mgm;
            un = (lt + eps * (cur.dl * mgm)) + fb;
            const double df = lt - un, dsq = df * df;
            const double dn2 = ((readlane_f64(dsq, 4) + readlane_f64(dsq, 20)) + readlane_f64(dsq, 36)) + readlane_f64(dsq, 52);
            dnan |= (dn2 != dn2);
            dmax = (dn2 > dmax) ? dn2 : dmax;                   // (sqrt is monotone: rooted once after the loop)
        }
        if (j == 4 && g < m) (uo + (size_t)t * m)[g] = un;
        xa = MFMA(BT, un * c4, xa);                             // + B u_t
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = fma(pb.kappa, xv[r] * xv[r] * xv[r], xa[r]) * c4;
        if (j == 4) {
            gbld *const xn = xo + (size_t)(t + 1) * n;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * r + g < n) xn[4 * r + g] = xv[r];
        }
    };
    int t0 = 0;
    for (; t0 + 4 <= N; t0 += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            issue(buf[(d + 3) & 3], t0 + d + 3);
            step(t0 + d, buf[d]);
        }
    }
    {
        const int nt = N - t0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < nt) step(t0 + d, buf[d]);
    }
    return dnan ? NAN : sqrt(dmax);
}
