// sweep_dual.h -- TWO independent Riccati recursions per wavefront over ONE pass through a trajectory's tiles.
//
//   recursion A : policy evaluation            solve_approximate_dp   (ileqg.jl:412-465)
//                   mode 6: initialize!'s open-loop sweep (L = 0, mu = 0)          (ileqg.jl:234)
//                   mode 7: line-search candidate 0 under the current gains         (ileqg.jl:522-528)
//   recursion B : the gain sweep the NEXT step! will run on these very tiles        solve_approximate_dp!  (ileqg.jl:341-406)
//                   (step! re-linearises the accepted trajectory, App. B.1: if candidate 0 is accepted -- or, in mode 6, if
//                   initialize! succeeds -- B's gains, mu, Delta are exactly what the reference computes next; otherwise they are
//                   discarded by ls_select_kernel / commit_init_kernel).  Its output goes to the idle half of the gain buffers.
//
// Why: at E = 1 a sweep kernel runs one wave per SIMD and that wave is idle ~45 % of its cycles waiting on its own serial
// pivot chain (profiles/r01_pmc_sq_E1.md).  Two waves per SIMD overlapped poorly; two recursions inside ONE wave give the
// compiler two independent dependency chains to interleave in a single basic block, and the tiles are read once for both.
// Every arithmetic expression is the one of sweep_kernel, so results are bit-identical to the unfused order (tested).
//
// B has no mu-restart loop here: if H is not PD (or M is not PD) B is abandoned (spec_st = 0 / 2) and A continues; the plain
// gain sweep then runs for that sample in the next round, restarts included.
#pragma once
#include "device_utils.h"
#include "kernels.h"
#include "layout.h"

// FLY sweeps (policy evaluation of the speculative path's line-search candidates, LQ family): the record of a candidate holds only its
// [c_x | c_u | c] row (rollin_body<.., NOTILE>); f_x | f_u and the cost Hessian of step t are formed here from x_t and the problem tables --
// the expressions of the rollout kernels' tile stores, so the registers hold the bits a materialised record would have delivered.
// FLY: 0 records are complete, 1 time-invariant cost tables (register images, loaded once), 2 time-varying cost tables (loaded per step)
struct FlyCtx {
    double zt[3], dg[3];   // register image of [A | B]; 1 on the lane that holds the diagonal element of f_x in that register
    d4 cc;                 // register image of [[Q, 0], [P, R]] (FLY == 1)
    double mq, kappa;
    const double *xh;      // state history of the trajectory
    const double *ctab;    // FLY == 2
};
__device__ __forceinline__ void fly_init(FlyCtx &fc, const ProblemDev &pb, const double *xh, int l, int g, int j, bool tv) {
    fc.mq = (j < 12) ? 1.0 : 0.0;
    fc.kappa = pb.kappa;
    fc.xh = xh;
    fc.ctab = pb.Ctab;
#pragma unroll
    for (int r = 0; r < 3; ++r) { fc.zt[r] = pb.Zt[64 * r + l]; fc.dg[r] = (j == 4 * r + g) ? 1.0 : 0.0; }
    fc.cc = (d4){0, 0, 0, 0};
    if (!tv) {
#pragma unroll
        for (int r = 0; r < 3; ++r) fc.cc[r] = pb.Ctab[64 * r + l] * fc.mq;
        fc.cc[3] = pb.Ctab[192 + l];
    }
}

struct DTile {
    d4 z, c;
    double x, la;          // x: lanes 0..15 = qr, lane 16 = q (register-image record, layout.h); la: own entry L[g][j] of the gain row block
    double xj;             // FLY: x_t[j] of this lane's column
};

// HASL: recursion A evaluates a given policy (mode 7); false for initialize!'s open-loop sweep (mode 6: all gains zero, nothing to load)
template <bool HASL, int FLY = 0>
__device__ __forceinline__ void dload(DTile &tr, const double *__restrict__ tp, int lx, int l, int j,
                                      const double *__restrict__ Lp, double mL, int g, const FlyCtx *fc = nullptr, int t = 0) {
    if (FLY) {
        tr.xj = fc->xh[(long)t * XSTR + ((j < 12) ? j : 11)];
        if (FLY == 2) {
            const double *__restrict__ C = fc->ctab + (long)t * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) tr.c[r] = C[64 * r + l];
        }
    } else {
        const double2 *__restrict__ t2 = reinterpret_cast<const double2 *>(tp);
        const int c34 = TS_REG(3, l), r5 = TS_REG(5, l);            // (loop-invariant per lane; dead lanes: the record's zero pair)
        const double2 w0 = t2[l], w1 = t2[64 + l], w2 = *reinterpret_cast<const double2 *>(tp + c34);
        tr.z[0] = w0.x; tr.z[1] = w0.y; tr.z[2] = w1.x; tr.z[3] = 0.0;
        tr.c[0] = w2.x; tr.c[1] = w2.y; tr.c[2] = tp[r5];
        tr.c[3] = w1.y;
    }
    tr.x = tp[TS_QR + lx];
    const int jc = (j < 12) ? j : 11;
    tr.la = HASL ? Lp[g * 12 + jc] * mL : 0.0;
}

// The elimination rounds of the two recursions are the shared elim_round (device_utils.h: rank-2 update on the matrix pipe, no LDS,
// no fence), issued back to back: two independent pivot chains for the scheduler to interleave.
// wls: this wavefront's LDS scratch (WLS_DUAL doubles)
template <int WM, bool HASL, int FLY = 0>
__device__ __forceinline__ void sweep_dual_body(const SweepArgs &a, const int b, double *const wls) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));      // opaque per phase (see sweep_body)
    const int l_ = lane_, g_ = l_ >> 4, j_ = l_ & 15;
    const int l = l_, g = g_, j = j_;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    [[maybe_unused]] const int dgs = 20 + 4 * (a.mode - 6);
    BODY_MARK(a.dump, dgs + 0);
    // (per-sample scalars fetched before the first test on any of them: see sweep_body)
    int slot, cidx = -1;
    const int v_act = st.ls_active[b], v_flag = st.flag_c[b * st.E], v_nom = st.slot_nom[b], v_stat = st.status[b], v_sel = st.lsel[b];
    const int s_act = wave_uniform(v_act), s_flag = wave_uniform(v_flag), s_nom = wave_uniform(v_nom), s_stat = wave_uniform(v_stat), sel = wave_uniform(v_sel);
    const double theta = st.theta[b], muB = st.mu[b];          // initialize! leaves mu = 0 (set by init_state_kernel)
    if (a.mode == 7) {
        if (!s_act) return;
        cidx = b * st.E;
        if (s_flag == 2) return;
        slot = cand_slot(b, 0, s_nom, st.E);
    } else {
        if (s_stat != ST_RUNNING) return;
        slot = b * (st.E + 1) + s_nom;
    }
    const double muA = (a.mode == 6) ? 0.0 : muB;
    const int N = st.N;
    if (a.mode == 7) BODY_MARK(a.dump, 28);
    const double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot) * st.tile_stride;
    FlyCtx fc;
    if (FLY) fly_init(fc, pb, st.xs + (long)slot * st.x_stride, l, g, j, FLY == 2);
    const double *__restrict__ Lb = st.L + (long)sel * st.l_half + (long)b * N * LSTR;
    double *__restrict__ Lout = st.L + (long)(sel ^ 1) * st.l_half + (long)b * N * LSTR;
    double *__restrict__ dlout = st.dl + (long)(sel ^ 1) * st.dl_half + (long)b * N * USTR;

    double *const lbufA = wls;
    double *const exA = wls + 64, *const exB = wls + 64 + 168;         // [G|H] 4x16, f 16, [80] = 0, [84..99] = s_vec, [104..] dump slots
                                                                       // of the idle lanes (unconditional writes: see sweep_body)
    if (l < 8) { exA[80 + l] = 0.0; exB[80 + l] = 0.0; }

    const double mL = (a.mode == 7 && j < 12) ? 1.0 : 0.0;
    const double m12 = (j < 12) ? 1.0 : 0.0;
    const double nth12 = -theta * m12;
    const double mA_ = (g == 0 && j < 12) ? 1.0 : 0.0, mB_ = (g == 0 && j == 12) ? 1.0 : 0.0, mH = (j == 12 + g) ? 1.0 : 0.0;
    ElimMasks em;
    elim_masks(em, g, j);
    int hoff[4], foff[3], goff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hoff[k] = (g <= k) ? (g * 16 + 12 + k) : (k * 16 + 12 + g);
        goff[k] = (j < 12) ? (k * 16 + j) : (j == 12 ? 64 + 12 + k : 80);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) foff[r] = (j == 12) ? (64 + 4 * r + g) : 80;
    const int gaoff = (j == 12) ? (64 + 12 + g) : 80;
    const int lx = (l < 17) ? l : TS_PAD - TS_QR;
    const int svo = (g == 0) ? 84 + j : 104 + l, fbo = (g == 0) ? 64 + j : 104 + l;
    double *const pgl = (j < 12) ? Lout + g * 12 + j : (j == 12 ? dlout + g : sample_sink(st, b) + l);
    const long sgl = (j < 12) ? LSTR : (j == 12 ? USTR : 0);

    d4 winv = {0, 0, 0, 0}, wp = {0, 0, 0, 0};
    double epall = 1.0;
    double nwrow[3] = {0.0, 0.0, 0.0};                           // WM == 2 (W diagonal): see sweep_body
    if (WM == 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r) nwrow[r] = -pb.Wdg[4 * r + g];
    }
    if (WM != 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { winv[r] = pb.Winv[64 * r + l]; wp[r] = pb.Wp[64 * r + l]; }
        epall = ((((pb.epiv[0] * pb.epiv[2]) * pb.epiv[4]) * pb.epiv[6]) * pb.epiv[8]) * pb.epiv[10];
    }
    const double coef = (theta != 0.0) ? -1.0 / (2.0 * theta) : 0.0;

    // terminal condition (ileqg.jl:352-354 / 429-431): both recursions start from the same V_N
    d4 vA, vB;
    {
        const double *__restrict__ tt = tile0 + (long)N * TSTRIDE;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = 4 * r + g;
            // (one unconditional load per register, the address selected per lane: conditional loads compile to a chain of divergent
            //  branches, each waiting for its own round trip -- 4.3 k cycles of prologue measured in the fused kernel)
            const double te = tt[(j < 12) ? TT_Q + i * 12 + j : TT_QV + i];
            vA[r] = (j <= 12) ? te : 0.0;
        }
        const double t3 = tt[(j < 12) ? TT_QV + j : TT_q];
        vA[3] = (g == 0 && j <= 12) ? (j < 12 ? t3 : 2.0 * t3) : 0.0;
        vB = vA;
    }
#ifdef RAT_DIAG_PHASES
    if (a.mode == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BODY_MARK(a.dump, 29); }
#endif
    double raccA = 0.0, raccB = 0.0, rprodA = 1.0, rprodB = 1.0;
    int rexpA = 0, rexpB = 0;
    int failA = 0, deadB = 0;        // deadB: 1 = H not PD (needs the restart loop of the plain kernel), 2 = M not PD

    DIAG_DECL
    auto step = [&](const int t, const DTile &cur) -> int {
        int l = l_, g = g_, j = j_;
        asm volatile("" : "+v"(l), "+v"(g), "+v"(j));
        DIAG_START();
        if (WM == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { winv[r] = pb.Winv[(long)t * 192 + 64 * r + l]; wp[r] = pb.Wp[(long)t * 192 + 64 * r + l]; }
            { const double *ept = pb.epiv + (long)t * 16; epall = ((((ept[0] * ept[2]) * ept[4]) * ept[6]) * ept[8]) * ept[10]; }
        }
        d4 cz, ccs;                                                  // the step's tile: from the record, or (FLY) formed here (see FlyCtx)
        if (!FLY) { cz = cur.z; ccs = cur.c; }
        else {
#pragma unroll
            for (int r = 0; r < 3; ++r) cz[r] = fx_diag(fc.zt[r], fc.dg[r], fc.kappa, cur.xj);
            cz[3] = 0.0;
            if (FLY == 2) { ccs[0] = cur.c[0] * fc.mq; ccs[1] = cur.c[1] * fc.mq; ccs[2] = cur.c[2] * fc.mq; ccs[3] = cur.c[3]; }
            else ccs = fc.cc;
        }
        d4 xzA, xzB;                                                 // (WM == 2: formed inside the theta == 0 branch, see sweep_body)
        if (WM != 2) {
            xzA = mm3(vA, cz, (d4){0, 0, 0, 0});
            xzB = mm3(vB, cz, (d4){0, 0, 0, 0});
        }
        d4 tmA, tmB;
        if (theta != 0.0) {
            exA[svo] = vA[3]; exB[svo] = vB[3];
            d4 mA, mB;
#pragma unroll
            for (int r = 0; r < 3; ++r) { mA[r] = fma(nth12, vA[r], winv[r]); mB[r] = fma(nth12, vB[r], winv[r]); }
            mA[3] = 0.0; mB[3] = 0.0;
            int pdA = 1, pdB = 1;
            rprodA *= epall; rprodB *= epall;
            DIAG_STAMP(0, mA[0]);
            elim_round_pair<0>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            elim_round_pair<1>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            elim_round_pair<2>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            elim_round_pair<3>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            elim_round_pair<4>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            elim_round_pair<5>(mA, mB, em, pdA, pdB, rprodA, rprodB);
            DIAG_STAMP(1, mA[0]);
            if (!(pdA > 0) || !(rprodA * 0.0 == 0.0)) { failA = 1; return 1; }    // @assert isposdef(M) (:440)
            if (!deadB && (!(pdB > 0) || !(rprodB * 0.0 == 0.0))) deadB = 2;      // @assert isposdef(M) (:366)
            rexpA += __builtin_amdgcn_frexp_exp(rprodA); rprodA = __builtin_amdgcn_frexp_mant(rprodA);
            rexpB += __builtin_amdgcn_frexp_exp(rprodB); rprodB = __builtin_amdgcn_frexp_mant(rprodB);
            if (WM == 2) {
                raccA += (nth12 * exA[84 + j]) * (mA[0] * exA[84 + g] + mA[1] * exA[84 + 4 + g] + mA[2] * exA[84 + 8 + g]);
                raccB += (nth12 * exB[84 + j]) * (mB[0] * exB[84 + g] + mB[1] * exB[84 + 4 + g] + mB[2] * exB[84 + 8 + g]);
                d4 mwA, mwB;
#pragma unroll
                for (int r = 0; r < 3; ++r) { mwA[r] = mA[r] * nwrow[r]; mwB[r] = mB[r] * nwrow[r]; }
                mwA[3] = 0.0; mwB[3] = 0.0;
                const d4 y2A = mm3(mwA, cz, (d4){0, 0, 0, 0});
                const d4 y2B = mm3(mwB, cz, (d4){0, 0, 0, 0});
                tmA = mm3(vA, y2A, (d4){0, 0, 0, 0});
                tmB = mm3(vB, y2B, (d4){0, 0, 0, 0});
            } else {
                d4 minvA, minvB;
#pragma unroll
                for (int r = 0; r < 3; ++r) { minvA[r] = nth12 * mA[r]; minvB[r] = nth12 * mB[r]; }
                minvA[3] = 0.0; minvB[3] = 0.0;
                raccA += exA[84 + j] * (minvA[0] * exA[84 + g] + minvA[1] * exA[84 + 4 + g] + minvA[2] * exA[84 + 8 + g]);
                raccB += exB[84 + j] * (minvB[0] * exB[84 + g] + minvB[1] * exB[84 + 4 + g] + minvB[2] * exB[84 + 8 + g]);
                const d4 y2A = mm3(minvA, xzA, (d4){0, 0, 0, 0});
                const d4 y2B = mm3(minvB, xzB, (d4){0, 0, 0, 0});
                tmA = mm3(vA, y2A, xzA);
                tmB = mm3(vB, y2B, xzB);
            }
        } else {
            // theta == 0: the reference still asserts isposdef(inv(W) - 0 S) (:365-366 / :439-440): a non-finite S fails it
            const double nfA = fma(vA[2], 0.0, fma(vA[1], 0.0, vA[0] * 0.0)), nfB = fma(vB[2], 0.0, fma(vB[1], 0.0, vB[0] * 0.0));
            if (__ballot(nfA != nfA) & 0x0FFF0FFF0FFF0FFFull) { failA = 1; return 1; }
            if (!deadB && (__ballot(nfB != nfB) & 0x0FFF0FFF0FFF0FFFull)) deadB = 2;
            raccA = fma(m12, fma(wp[2], vA[2], fma(wp[1], vA[1], wp[0] * vA[0])), raccA);
            raccB = fma(m12, fma(wp[2], vB[2], fma(wp[1], vB[1], wp[0] * vB[0])), raccB);
            if (WM == 2) { tmA = mm3(vA, cz, (d4){0, 0, 0, 0}); tmB = mm3(vB, cz, (d4){0, 0, 0, 0}); }
            else { tmA = xzA; tmB = xzB; }
        }
        DIAG_STAMP(2, tmA[0]);
        d4 fA = mm3(cz, tmA, ccs);
        d4 fB = mm3(cz, tmB, ccs);
        const double ghA = fma(muA, mH, fA[3]), ghB = fma(muB, mH, fB[3]);
        const double fvA = tmA[3] + cur.x, fvB = tmB[3] + cur.x;
        exA[g * 16 + j] = ghA; exB[g * 16 + j] = ghB;
        exA[fbo] = fvA; exB[fbo] = fvB;
        if (HASL) lbufA[l] = cur.la;                            // rows of [L | dl] of the given policy to every lane
        DIAG_STAMP(3, ghB);
        WAVE_SYNC();
        // ---- A: given policy (:446-451) ----
        const double hA0 = exA[hoff[0]], hA1 = exA[hoff[1]], hA2 = exA[hoff[2]], hA3 = exA[hoff[3]];
        const double gaA = fma(ghA, m12, exA[gaoff]);
        const double uaA = HASL ? hA0 * lbufA[j] + hA1 * lbufA[16 + j] + hA2 * lbufA[32 + j] + hA3 * lbufA[48 + j] + gaA : gaA;
        // ---- B: optimal gains (:372-382) ----
        const double hB0 = exB[hoff[0]], hB1 = exB[hoff[1]], hB2 = exB[hoff[2]], hB3 = exB[hoff[3]];
        const double gaB = fma(ghB, m12, exB[gaoff]);
        const double h00 = exB[12], h01 = exB[13], h02 = exB[14], h03 = exB[15];
        const double h11 = exB[16 + 13], h12 = exB[16 + 14], h13 = exB[16 + 15];
        const double h22 = exB[32 + 14], h23 = exB[32 + 15], h33 = exB[48 + 15];
        const double g0 = exB[goff[0]], g1 = exB[goff[1]], g2 = exB[goff[2]], g3 = exB[goff[3]];
        const double d0 = h00, i0 = fast_rcp(d0);
        const double l10 = h01 * i0, l20 = h02 * i0, l30 = h03 * i0;
        const double d1 = h11 - l10 * h01, i1 = fast_rcp(d1);
        const double l21 = (h12 - l20 * h01) * i1, l31 = (h13 - l30 * h01) * i1;
        const double d2 = h22 - l20 * h02 - l21 * (l21 * d1), i2 = fast_rcp(d2);
        const double l32 = (h23 - l30 * h02 - l31 * (l21 * d1)) * i2;
        const double d3 = h33 - l30 * h03 - l31 * (l31 * d1) - l32 * (l32 * d2), i3 = fast_rcp(d3);
        if (!deadB && !(d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) deadB = 1;    // !isposdef(H) (:372): left to the plain kernel
        const double y0 = -g0;
        const double y1 = -g1 - l10 * y0;
        const double y2 = -g2 - l20 * y0 - l21 * y1;
        const double y3 = -g3 - l30 * y0 - l31 * y1 - l32 * y2;
        const double x3 = y3 * i3;
        const double x2 = y2 * i2 - l32 * x3;
        const double x1 = y1 * i1 - l21 * x2 - l31 * x3;
        const double x0 = y0 * i0 - l10 * x1 - l20 * x2 - l30 * x3;
        const double laB = ((x0 * em.e0[0] + x1 * em.e1[0]) + x2 * em.e0[1]) + x3 * em.e1[1];     // row g takes x_g (see sweep_body)
        const double uaB = hB0 * x0 + hB1 * x1 + hB2 * x2 + hB3 * x3 + gaB;
        pgl[(long)t * sgl] = (j <= 12) ? laB : 0.0;                // L_t | dl_t | idle lanes: sink
        d4 fxA, fxB;
#pragma unroll
        for (int r = 0; r < 3; ++r) { fxA[r] = fma(fA[r], m12, exA[foff[r]]); fxB[r] = fma(fB[r], m12, exB[foff[r]]); }
        const double qc = readlane_f64(cur.x, 16);
        fxA[3] = fma(fvA, mA_, (2.0 * qc + vA[3]) * mB_);
        fxB[3] = fma(fvB, mA_, (2.0 * qc + vB[3]) * mB_);
        DIAG_STAMP(4, uaB);
        d4 vnB = MFMA(laB, uaB, fxB);
        vnB = MFMA(gaB, laB, vnB);
        if (HASL) {
            d4 vnA = MFMA(cur.la, uaA, fxA);
            vA = MFMA(gaA, cur.la, vnA);
        } else {
            vA = fxA;                                           // zero gains: V = Fx
        }
        vB = vnB;
        DIAG_STAMP(5, vB[0]);
        WAVE_SYNC();
        return 0;
    };

    if (a.mode == 7) BODY_MARK(a.dump, 30);
    DTile ra, rb2;
    dload<HASL, FLY>(ra, tile0 + (long)(N - 1) * TSTRIDE, lx, l, j, Lb + (long)(N - 1) * LSTR, mL, g, &fc, N - 1);
    BODY_MARK(a.dump, dgs + 1);
    for (int t = N - 1; t >= 0; t -= 2) {
        {
            const int tn = (t > 0) ? t - 1 : 0;
            dload<HASL, FLY>(rb2, tile0 + (long)tn * TSTRIDE, lx, l, j, Lb + (long)tn * LSTR, mL, g, &fc, tn);
        }
        if (step(t, ra)) break;
        if (t == 0) break;
        {
            const int tn = (t > 1) ? t - 2 : 0;
            dload<HASL, FLY>(ra, tile0 + (long)tn * TSTRIDE, lx, l, j, Lb + (long)tn * LSTR, mL, g, &fc, tn);
        }
        if (step(t - 1, rb2)) break;
    }
    BODY_MARK(a.dump, dgs + 2);
#ifdef RAT_DIAG
    if (l_ == 0 && blockIdx.x < 8 && a.dump)
        for (int q = 0; q < 6; ++q) a.dump[128 + blockIdx.x * 8 + q] = (double)dg_acc[q];
#endif
    const double totA = sweep_scalars(wave_sum(raccA), coef, rprodA, rexpA, theta != 0.0);
    (void)raccB; (void)rprodB; (void)rexpB;          // B's value s_1 is not used by step! (only L, dl, mu, Delta are)
    if (l == 12) {
        const double s0 = 0.5 * vA[3] + totA;
        if (a.mode == 7) {
            st.value_c[cidx] = s0;
            st.flag_c[cidx] = failA ? 1 : 0;
            if (a.prune) {
                // Will line_search! settle on this candidate (ileqg.jl:538, or the forced accept of :557-558)?  Then the sequential rule never
                // reads candidates 1 .. E-1 of this round: their evaluation waves poll this word and stop (sweep_body<.., PRUNE>).
                const double cur = st.value[b], eps = st.ls_eps[b];
                const bool take = !failA && (isapprox_default(s0, cur) || s0 < cur || eps * a.op.lambda < a.op.eps_min);
                __atomic_store_n(&st.acc0[b], take ? 1 : 0, __ATOMIC_RELAXED);
            }
        } else {
            st.value[b] = failA ? INFINITY : s0;
            if (failA) st.status[b] = 1;                      // RAT_ST_M_NOT_PD_INIT
        }
        st.mu_spec[b] = muB;                                   // no restart happened: mu, Delta unchanged by the gain sweep
        st.delta_spec[b] = st.delta[b];
        st.spec_st[b] = (failA || deadB == 1) ? 0 : (deadB == 2 ? 2 : 1);
    }
    BODY_MARK(a.dump, dgs + 3);
}

