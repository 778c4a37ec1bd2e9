// psweep.h -- TIME-PARALLEL (segment-parallel) risk-sensitive Riccati sweep: solve_approximate_dp(!) of ONE trajectory by the P wavefronts
// of a workgroup (ileqg.jl:341-406 / 412-465).  tests/psweep_model.py is the NumPy model of this file; DESIGN.md section 3 the account.
//
// The backward recursion is a chain of N dependent steps (~3,000-3,600 cycles each) that one in-order wavefront walks alone while the
// sample's other SIMDs idle.  Here the horizon is cut into P segments [cut_s, cut_s+1):
//   wave P-1          runs the recursion itself on the last segment, from the terminal condition (and carries on through the segment before);
//   the other waves   meanwhile build the ELEMENT of a segment each -- the map (S, s_vec) at its end -> (S, s_vec) at its start, in the
//                     conditional-value-function form of the associative LQ scan (Sarkka & Garcia-Fernandez) -- with no knowledge of the
//                     value function at the segment's end: the recursion from a ZERO terminal value (Jv), the transpose K of the
//                     segment's closed-loop transition in homogeneous coordinates, the accumulated noise covariance (-Sigbar) and, for
//                     gain sweeps, the accumulated control authority Ubar.  Prepending a step costs the step itself + 10 (11) MFMAs: the
//                     inverse M^-1 = (inv(W) - theta Jv)^-1 of the step is the only inversion;
//   the chain         the true value at a boundary is handed upstream through LDS; the wave that owns the element applies it ("hop", in
//                     information form: Xt = (S_b^-1 + Ubar - theta Sigbar)^-1 by two SPD eliminations, V = Jv + Aa' Xt Aa, 6 MFMAs);
//   phase 3           with its true boundary value every wave re-runs the ORDINARY step over its own segment: gains, isposdef tests,
//                     logdet(W M), theta s'M^-1 s and the additive scalar come from the sequential sweep's own arithmetic; only the
//                     boundary values differ from the sequential sweep's, by rounding (measured <= 2e-15 relative).
// P waves cover P + 1 segments: wave w <= P-2 builds the element of segment w+1, hops, and runs phase 3 of segment w; wave P-1 runs the
// recursion over segments P and P-1.  Critical path: a + (P - 1) hops + b steps instead of N (a: last segment, b: first) -- measured
// (N = 50, P = 4): an element step costs 1.23 ordinary steps, a hop 1.1 - 1.4.  Anything the element form cannot decide -- theta
// == 0 (its own recursion), a non-positive pivot while building an element or inside a hop (S_b singular: no state cost), NaNs -- makes
// wave 0 run the sequential sweep_body instead: results never depend on the shortcut being available.
#pragma once
#include <type_traits>

#define WLS_PSW (WLS_SWEEP + 256 + 16)     /* per wave: sweep_body's scratch + a 16 x 16 transposition pad + 16 doubles (w) */

// (PswCuts: kernels.h -- cut[0] = 0 < cut[1] < ... < cut[P] = N)

template <int MP>                          // MP: the largest team this area serves
struct PswSharedT {
    double vbox[MP][256];                  // vbox[s]: the true value at cut[s+1] (accumulator-layout image), written by wave s+1
    double part[MP][4];                    // per wave: 0.5 V[12][12], sum of racc, log term, -
    int flag[MP];                          // flag[s] = epoch of the attempt once vbox[s] is valid
    int bar;                               // team barrier: arrivals so far (a multiple of P between calls); zero before the first call
    int fail_def;                          // a definite M-not-PD (the ordinary step on true values)
    int uncertain;                         // the element form could not decide: sequential fallback
    int hnotpd;                            // gain sweeps: H not PD in an ordinary step (mu restart)
    int last_rc;                           // (attempt << 2) | what ended the LAST segment's recursion early (1 M not PD, 2 H not PD): definite,
                                           // the sequential sweep meets it first; the other waves skip the rest of the attempt
    int lastP;                             // diagnostic builds: the team size of the previous call on this area (0: none yet)
};
using PswShared = PswSharedT<PSW_MAXP>;

__device__ __forceinline__ void psw_spin(int *const word, const int want) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - want < 0) __builtin_amdgcn_s_sleep(1);
    WAVE_SYNC();
}

// all four 16-lane rows summed, on every lane: ((r0 + r1) + r2) + r3
__device__ __forceinline__ double rows_sum4(double x) {
    double r[4];
    rows_bcast(x, r);
    return ((r[0] + r[1]) + r[2]) + r[3];
}

// phase-timeline build only (make diagp; tools/psweep_phases.py): cycle stamps per wave at the phase boundaries
#ifdef RAT_DIAG_PHASES
#define PSW_MARK(slot_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8 && a.dump) \
        a.dump[4096 + blockIdx.x * 512 + (a.mode & 7) * 64 + wave * 16 + (slot_)] = (double)__builtin_readcyclecounter(); } while (0)
#else
#define PSW_MARK(slot_) do {} while (0)
#endif

template <bool GAIN, int WM, bool HASL, int FLY, class SH>
__device__ __forceinline__ void psweep_body(const SweepArgs &a, const int tid, double *const wls, SH *const sh, const PswCuts &pc, const int wave) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l_ = lane_, g_ = l_ >> 4, j_ = l_ & 15;
    const int l = l_, g = g_, j = j_;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int P = pc.P;
    if (wave >= P && wave > 0) return;             // (a horizon too short for the team: the spare waves sit the sweep out)
    int b, k = 0, slot, cidx = -1;
    if (a.mode == 1) { const int Ek = st.E - a.k_first; b = tid / Ek; k = a.k_first + (tid - b * Ek); } else b = tid;
    const int fidx = b * st.E + k;
    const int v_act = xld(&st.ls_active[b]), v_flag = xld(&st.flag_c[fidx]), v_nom = xld(&st.slot_nom[b]), v_stat = xld(&st.status[b]), v_sel = xld(&st.lsel[b]);
    const int s_act = wave_uniform(v_act), s_flag = wave_uniform(v_flag), s_nom = wave_uniform(v_nom), s_stat = wave_uniform(v_stat), sel = wave_uniform(v_sel);
    const double theta = xld(&st.theta[b]), mu_in = xld(&st.mu[b]);
    double delta = xld(&st.delta[b]);
    // (the same words decide for every wave of the workgroup: they leave together)
    if (a.mode == 1) {
        if (!s_act) return;
        cidx = fidx;
        if (s_flag == 2) return;
        slot = cand_slot(b, k, s_nom, st.E);
    } else if (a.mode == 4) {
        if (!s_act) return;
        if (s_flag == 2) return;
        slot = cand_slot(b, 0, s_nom, st.E);
    } else if (a.mode == 5) {
        slot = b * (st.E + 1) + s_nom;
    } else {
        if (s_stat != ST_RUNNING) return;
        if (a.mode == 0 && s_act) return;
        slot = b * (st.E + 1) + s_nom;
    }
    // what the element form does not cover runs sequentially on wave 0 (sweep_body writes every output itself)
    const bool theta0 = wave_uniform((theta == 0.0 || !(theta == theta)) ? 1 : 0) != 0;
    if (theta0 || P < 2) {
        if (wave == 0) sweep_body<GAIN, false, WM, HASL, 0, FLY>(a, tid, wls);
        return;
    }
    double mu = (a.mode == 2) ? 0.0 : mu_in;
    const int N = st.N;
    const double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot) * st.tile_stride;
    FlyCtx fc;
    if (FLY) fly_init(fc, pb, st.xs + (long)slot * st.x_stride, l, g, j, FLY == 2);
    const int osel = (a.mode >= 4) ? (sel ^ 1) : sel;
    const double *__restrict__ Lb = st.L + (long)sel * st.l_half + (long)b * N * LSTR;
    double *__restrict__ Lout = st.L + (long)osel * st.l_half + (long)b * N * LSTR;
    double *__restrict__ dlout = st.dl + (long)osel * st.dl_half + (long)b * N * USTR;

    double *const lbuf = wls;
    double *const ex = wls + 64;
    double *const tpad = wls + WLS_SWEEP;          // 16 x 16 pad: transposition of K, the element's B'T_K rows
    double *const wpad = tpad + 256;               // 16: w = S_b^-1 s_b by component
#define HBUF(r_, c_) ex[(r_) * 16 + (c_)]
#define FBUF(c_) ex[64 + (c_)]
#define SVB(c_) ex[84 + (c_)]
    if (l < 8) ex[80 + l] = 0.0;

    const double mL = ((a.mode == 1) && j < 12) ? 1.0 : 0.0;
    const double m12 = (j < 12) ? 1.0 : 0.0;
    const double nth12 = -theta * m12;
    const double mA = (g == 0 && j < 12) ? 1.0 : 0.0, mB = (g == 0 && j == 12) ? 1.0 : 0.0, mH = (j == 12 + g) ? 1.0 : 0.0;
    ElimMasks em;
    elim_masks(em, g, j);
    int hoff[4], foff[3], goff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hoff[q] = (g <= q) ? (g * 16 + 12 + q) : (q * 16 + 12 + g);
        goff[q] = (j < 12) ? (q * 16 + j) : (j == 12 ? 64 + 12 + q : 80);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) foff[r] = (j == 12) ? (64 + 4 * r + g) : 80;
    const int gaoff = (j == 12) ? (64 + 12 + g) : 80;
    const int svo = (g == 0) ? 84 + j : 104 + l, fbo = (g == 0) ? 64 + j : 104 + l;
    double *const pgl = (j < 12) ? Lout + g * 12 + j : (j == 12 ? dlout + g : sample_sink(st, b) + l);
    const long sgl = (j < 12) ? LSTR : (j == 12 ? USTR : 0);
    const int lx = (l < 17) ? l : TS_PAD - TS_QR;
    const int lq = g * 12 + ((j < 12) ? j : 11);

    d4 winv = {0, 0, 0, 0}, wp = {0, 0, 0, 0};
    double epall = 1.0;
    double nwrow[3] = {0.0, 0.0, 0.0};
    if (WM == 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r) nwrow[r] = -pb.Wdg[4 * r + g];
    }
    if (WM != 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            winv[r] = pb.Winv[64 * r + l];
            wp[r] = pb.Wp[64 * r + l];
        }
        epall = ((((pb.epiv[0] * pb.epiv[2]) * pb.epiv[4]) * pb.epiv[6]) * pb.epiv[8]) * pb.epiv[10];
    }
    const double coef = -1.0 / (2.0 * theta);
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};

    int restarts = 0;
    int fail = 0;                  // 1: M not PD (definite), 5: mu diverged
    d4 v = zero4;
    d4 kk = zero4, nsig = zero4, ub = zero4;      // the element: K, -Sigbar, Ubar
    double racc = 0.0, rprod = 1.0;
    int rexp = 0;

    // one backward step on the tile registers `cur`.  COMP: the step also prepends itself to the element (kk, nsig, ub).
    // returns 0, 1 (M not PD), 2 (H not PD)
    auto step = [&](auto comp_tag, const int t, const TileRegs &cur) -> int {
        constexpr bool COMP = decltype(comp_tag)::value;
        int l = l_, g = g_, j = j_;
        asm volatile("" : "+v"(l), "+v"(g), "+v"(j));
        if (WM == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                winv[r] = pb.Winv[(long)t * 192 + 64 * r + l];
                wp[r] = pb.Wp[(long)t * 192 + 64 * r + l];
            }
            const double *ept = pb.epiv + (long)t * 16;
            epall = ((((ept[0] * ept[2]) * ept[4]) * ept[6]) * ept[8]) * ept[10];
        }
        d4 cz, ccs;
        if (!FLY) { cz = cur.z; ccs = cur.c; }
        else {
#pragma unroll
            for (int r = 0; r < 3; ++r) cz[r] = fx_diag(fc.zt[r], fc.dg[r], fc.kappa, cur.xj);
            cz[3] = 0.0;
            if (FLY == 2) { ccs[0] = cur.c[0] * fc.mq; ccs[1] = cur.c[1] * fc.mq; ccs[2] = cur.c[2] * fc.mq; ccs[3] = cur.c[3]; }
            else ccs = fc.cc;
        }
        d4 xz;
        if (WM != 2) xz = mm3(v, cz, zero4);
        if (HASL) lbuf[l] = cur.la;
        const double qc = readlane_f64(cur.x, 16);
        d4 tm;
        ex[svo] = v[3];
        d4 m;
#pragma unroll
        for (int r = 0; r < 3; ++r) m[r] = fma(nth12, v[r], winv[r]);
        m[3] = 0.0;
        int pdmin = 1;
        rprod *= epall;
        elim_round<0, 0>(m, em, pdmin, rprod);
        elim_round<1, 0>(m, em, pdmin, rprod);
        elim_round<2, 0>(m, em, pdmin, rprod);
        elim_round<3, 0>(m, em, pdmin, rprod);
        elim_round<4, 0>(m, em, pdmin, rprod);
        elim_round<5, 0>(m, em, pdmin, rprod);
        if (!(pdmin > 0) || !(rprod * 0.0 == 0.0)) return 1;
        rexp += __builtin_amdgcn_frexp_exp(rprod);
        rprod = __builtin_amdgcn_frexp_mant(rprod);
        // ---- element: G = M^-1 K[:12] (gneg = -G), T_K = [inv(W) G ; K[12] + theta s_vec' G] ------------------------------------------
        d4 gneg = zero4, tk = zero4;
        double tk3 = 0.0;
        if (COMP) {
            gneg = mm3(m, kk, zero4);                                          // m = -M^-1 (symmetric)
            const double part = (SVB(g) * gneg[0] + SVB(4 + g) * gneg[1]) + SVB(8 + g) * gneg[2];      // -(s_vec' G), this row group's share
            const double ssum = rows_sum4(part);
            tk3 = fma(-theta, ssum, kk[3]) * (mA + mB);                 // (row 12 of the image: register 3, first 16-lane row, columns <= 12)
            if (WM == 2) {
#pragma unroll
                for (int r = 0; r < 3; ++r) tk[r] = gneg[r] * nwrow[r];      // (-G)(-inv(W)_ii): inv(W) G, W diagonal
            } else {
                d4 gp;
#pragma unroll
                for (int r = 0; r < 3; ++r) gp[r] = -gneg[r];
                gp[3] = 0.0;
                tk = mm3(winv, gp, zero4);                                     // inv(W) G (inv(W) symmetric)
            }
            tk[3] = 0.0;
        }
        if (WM == 2) {
            racc += (nth12 * SVB(j)) * (m[0] * SVB(g) + m[1] * SVB(4 + g) + m[2] * SVB(8 + g));
            d4 mw;
#pragma unroll
            for (int r = 0; r < 3; ++r) mw[r] = m[r] * nwrow[r];
            mw[3] = 0.0;
            const d4 y2 = mm3(mw, cz, zero4);
            tm = mm3(v, y2, zero4);
        } else {
            d4 minv;
#pragma unroll
            for (int r = 0; r < 3; ++r) minv[r] = nth12 * m[r];
            minv[3] = 0.0;
            racc += SVB(j) * (minv[0] * SVB(g) + minv[1] * SVB(4 + g) + minv[2] * SVB(8 + g));
            const d4 y2 = mm3(minv, xz, zero4);
            tm = mm3(v, y2, xz);
        }
        d4 f = mm3(cz, tm, ccs);
        d4 fk = zero4;
        if (COMP) {
            fk = mm3(cz, tk, zero4);                                           // rows 0..11: A' T_K, register 3: B' T_K (natural 4 x 16 rows)
            nsig = mm3(kk, gneg, nsig);                                        // -Sigbar -= K[:12]' G
        }
        const double gh = fma(mu, mH, f[3]);
        const double fv = tm[3] + cur.x;
        HBUF(g, j) = gh;
        ex[fbo] = fv;
        if (COMP && GAIN) tpad[l] = fk[3];
        WAVE_SYNC();
        const double hg0 = ex[hoff[0]], hg1 = ex[hoff[1]], hg2 = ex[hoff[2]], hg3 = ex[hoff[3]];
        const double ga = fma(gh, m12, ex[gaoff]);
        double x0, x1, x2, x3, la;
        if (GAIN) {
            const double h00 = HBUF(0, 12), h01 = HBUF(0, 13), h02 = HBUF(0, 14), h03 = HBUF(0, 15);
            const double h11 = HBUF(1, 13), h12 = HBUF(1, 14), h13 = HBUF(1, 15);
            const double h22 = HBUF(2, 14), h23 = HBUF(2, 15), h33 = HBUF(3, 15);
            const double g0 = ex[goff[0]], g1 = ex[goff[1]], g2 = ex[goff[2]], g3 = ex[goff[3]];
            const double d0 = h00, i0 = fast_rcp(d0);
            const double l10 = h01 * i0, l20 = h02 * i0, l30 = h03 * i0;
            const double d1 = h11 - l10 * h01, i1 = fast_rcp(d1);
            const double l21 = (h12 - l20 * h01) * i1, l31 = (h13 - l30 * h01) * i1;
            const double d2 = h22 - l20 * h02 - l21 * (l21 * d1), i2 = fast_rcp(d2);
            const double l32 = (h23 - l30 * h02 - l31 * (l21 * d1)) * i2;
            const double d3 = h33 - l30 * h03 - l31 * (l31 * d1) - l32 * (l32 * d2), i3 = fast_rcp(d3);
            if (!(d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) return 2;
            const double y0 = -g0;
            const double y1 = -g1 - l10 * y0;
            const double y2 = -g2 - l20 * y0 - l21 * y1;
            const double y3 = -g3 - l30 * y0 - l31 * y1 - l32 * y2;
            x3 = y3 * i3;
            x2 = y2 * i2 - l32 * x3;
            x1 = y1 * i1 - l21 * x2 - l31 * x3;
            x0 = y0 * i0 - l10 * x1 - l20 * x2 - l30 * x3;
            la = ((x0 * em.e0[0] + x1 * em.e1[0]) + x2 * em.e0[1]) + x3 * em.e1[1];
            if (COMP) {
                // Ubar += (B'T_K)' H^-1 (B'T_K): column j of B'T_K solved with the step's own factors, each lane its column
                const double c0 = tpad[j], c1 = tpad[16 + j], c2 = tpad[32 + j], c3 = tpad[48 + j];
                const double z0 = c0;
                const double z1 = c1 - l10 * z0;
                const double z2 = c2 - l20 * z0 - l21 * z1;
                const double z3 = c3 - l30 * z0 - l31 * z1 - l32 * z2;
                const double w3 = z3 * i3;
                const double w2 = z2 * i2 - l32 * w3;
                const double w1 = z1 * i1 - l21 * w2 - l31 * w3;
                const double w0 = z0 * i0 - l10 * w1 - l20 * w2 - l30 * w3;
                const double xk = ((w0 * em.e0[0] + w1 * em.e1[0]) + w2 * em.e0[1]) + w3 * em.e1[1];
                ub = MFMA(fk[3], xk, ub);
            }
        } else if (HASL) {
            x0 = lbuf[j]; x1 = lbuf[16 + j]; x2 = lbuf[32 + j]; x3 = lbuf[48 + j];
            la = cur.la;
        } else {
            x0 = x1 = x2 = x3 = la = 0.0;
        }
        const double ua = hg0 * x0 + hg1 * x1 + hg2 * x2 + hg3 * x3 + ga;
        if (GAIN) pgl[(long)t * sgl] = (j <= 12) ? la : 0.0;
        if (COMP) {
            // K <- [A' T_K ; T_K[12]] + La' (B' T_K)   (A_i' T_K with the closed-loop transition of the step)
            const d4 kbase = {fk[0], fk[1], fk[2], tk3};
            if (GAIN || HASL) kk = MFMA(la, fk[3], kbase); else kk = kbase;
        }
        d4 fx;
#pragma unroll
        for (int r = 0; r < 3; ++r) fx[r] = fma(f[r], m12, ex[foff[r]]);
        fx[3] = fma(fv, mA, (2.0 * qc + v[3]) * mB);
        if (GAIN || HASL) {
            d4 vn = MFMA(la, ua, fx);
            vn = MFMA(ga, la, vn);
            v = vn;
        } else {
            v = fx;
        }
        WAVE_SYNC();
        return 0;
    };
    // steps thi-1 .. tlo over ping-pong tile registers (prefetch clamped at the segment's first step)
    auto run = [&](auto comp_tag, const int thi, const int tlo) -> int {
        TileRegs nx, rb;
        // (every memory operation issued so far lands before the loop is entered: a load still in flight at the loop's entry makes the
        //  compiler's wait at the loop HEADER a vmcnt(0) -- paid on every step, it drains the gain stores and the prefetch: +500 cycles a
        //  step; so does any branch inside the loop, which is why the last wave's hand-over sits between two calls, not inside one)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        load_tile<HASL, false, FLY>(nx, tile0 + (long)(thi - 1) * TSTRIDE, l, lx, lq, Lb + (long)(thi - 1) * LSTR, nullptr, mL, g, j, &fc, thi - 1);
        for (int t = thi - 1; t >= tlo; t -= 2) {
            {
                const int tn = (t > tlo) ? t - 1 : tlo;
                load_tile<HASL, false, FLY>(rb, tile0 + (long)tn * TSTRIDE, l, lx, lq, Lb + (long)tn * LSTR, nullptr, mL, g, j, &fc, tn);
            }
            if (const int r = step(comp_tag, t, nx)) return r;
            if (t == tlo) break;
            {
                const int tn = (t > tlo + 1) ? t - 2 : tlo;
                load_tile<HASL, false, FLY>(nx, tile0 + (long)tn * TSTRIDE, l, lx, lq, Lb + (long)tn * LSTR, nullptr, mL, g, j, &fc, tn);
            }
            if (const int r = step(comp_tag, t - 1, rb)) return r;
        }
        return 0;
    };
    using TrueTag = std::integral_constant<bool, true>;
    using FalseTag = std::integral_constant<bool, false>;

    // Team barrier of the P waves (LDS counter; no s_barrier: inside solve_block_kernel two teams run different sweeps side by side).
    // sh->bar is a multiple of P whenever no wave of the team is inside this body: a PswShared serves teams of ONE size.
    int gen = wave_uniform(__hip_atomic_load(&sh->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) / P;
#ifdef RAT_DIAG_PHASES
    // diagnostic build: the invariant's sufficient condition, checked -- a barrier area serves teams of ONE size (a four-wave call after
    // two-wave calls on the same counter released early and summed stale partials in round 5).  A violation is counted
    // in the dump area: tools/gpu_phases_bpsw.py / gpu_phases_duo.py print the count.
    if (wave == 0 && l_ == 0) {
        if (sh->lastP != 0 && sh->lastP != P && a.dump && a.st.N >= 36) atomicAdd(&a.dump[4095], 1.0);
        sh->lastP = P;
    }
#endif
    auto team_barrier = [&]() {
        WAVE_SYNC();
        ++gen;
        if (l_ == 0) __hip_atomic_fetch_add(&sh->bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        psw_spin(&sh->bar, gen * P);
    };
    auto post = [&](const d4 &val, const int box, const int epoch) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sh->vbox[box][64 * q + l] = val[q];
        WAVE_SYNC();
        if (l == 0) __hip_atomic_store(&sh->flag[box], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // P waves, P + 1 segments [cut[s], cut[s+1]):
    //   wave P-1   the recursion itself over segment P from the terminal condition, posts the value at cut[P], and carries on through
    //              segment P-1;
    //   wave w     (w <= P-2) builds the element of segment w+1 meanwhile, takes the true value at cut[w+2] from wave w+1, hops to
    //              cut[w+1], posts that for wave w-1, and runs the ordinary recursion over segment w from it.
    while (true) {
        if (wave == 0 && l == 0) { sh->fail_def = 0; sh->uncertain = 0; sh->hnotpd = 0; }
        team_barrier();
        const int epoch = gen;                       // unique per attempt and per call: what the flags of this attempt carry
        int my_fail = 0, my_unc = 0, my_h = 0;
        PSW_MARK(0);
        bool dead = false;
        int thi, tlo, tmid;                          // the ordinary recursion runs over [tmid, thi) and then [tlo, tmid)
        if (wave == P - 1) {
            // terminal condition (ileqg.jl:352-354 / 429-431)
            const double *__restrict__ tt = tile0 + (long)N * TSTRIDE;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = 4 * r + g;
                const double te = xld(&tt[(j < 12) ? TT_Q + i * 12 + j : TT_QV + i]);
                v[r] = (j <= 12) ? te : 0.0;
            }
            const double t3 = xld(&tt[(j < 12) ? TT_QV + j : TT_q]);
            v[3] = (g == 0 && j <= 12) ? (j < 12 ? t3 : 2.0 * t3) : 0.0;
            thi = N; tlo = pc.cut[P - 1]; tmid = pc.cut[P];
        } else {
            // ---- phase 1: the element of segment wave + 1 ---------------------------------------------------------------------------------------
            d4 ac = zero4;
            v = zero4;
            kk = (d4){(j == g) ? 1.0 : 0.0, (j == 4 + g) ? 1.0 : 0.0, (j == 8 + g) ? 1.0 : 0.0, (g == 0 && j == 12) ? 1.0 : 0.0};
            nsig = zero4; ub = zero4;
            racc = 0.0; rprod = 1.0; rexp = 0;
            {
                const int r = run(TrueTag(), pc.cut[wave + 2], pc.cut[wave + 1]);
                if (r) my_unc = 1;
            }
            // A_c = K' through the pad (once per segment, off the chain)
#pragma unroll
            for (int q = 0; q < 4; ++q) tpad[(4 * q + g) * 16 + j] = kk[q];
            WAVE_SYNC();
#pragma unroll
            for (int q = 0; q < 3; ++q) ac[q] = tpad[j * 16 + 4 * q + g];
            WAVE_SYNC();
            PSW_MARK(1);
            // ---- the true value at the end of that segment ------------------------------------------------------------------------------------
            psw_spin(&sh->flag[wave], epoch);
            PSW_MARK(2);
            dead = (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sh->last_rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 2) == epoch;
            d4 vt = zero4;
            if (!dead) {
                // ---- hop: V at the segment's start = Jv + Aa' (S_b^-1 + Ubar - theta Sigbar)^-1 Aa,  Aa = [Abar | bbar + S_b^-1 s_b] ----------
                int l = l_, g = g_, j = j_;
                asm volatile("" : "+v"(l), "+v"(g), "+v"(j));      // (the hop's lane masks are formed here, not kept live across the time loops)
                const double mcol12 = (j == 12) ? 1.0 : 0.0;
                d4 vb;
#pragma unroll
                for (int q = 0; q < 3; ++q) vb[q] = sh->vbox[wave][64 * q + l];
                const double s0 = sh->vbox[wave][192 + g], s1 = sh->vbox[wave][192 + 4 + g], s2 = sh->vbox[wave][192 + 8 + g];   // s_b by row index
                // padded coordinates (n < 12): 1 on their diagonal for the inversions; the padding's unit "noise" (inv(W) is padded with 1) is
                // cleared from Cbar there
                d4 sb;
#pragma unroll
                for (int r = 0; r < 3; ++r) sb[r] = fma(vb[r], m12, (4 * r + g == j && j >= pb.n) ? 1.0 : 0.0);
                sb[3] = 0.0;
                int pd = 1;
                double rp = 1.0;
                elim_round<0, 0>(sb, em, pd, rp);
                elim_round<1, 0>(sb, em, pd, rp);
                elim_round<2, 0>(sb, em, pd, rp);
                elim_round<3, 0>(sb, em, pd, rp);
                elim_round<4, 0>(sb, em, pd, rp);
                elim_round<5, 0>(sb, em, pd, rp);                                  // sb = -S_b^-1
                if (!(pd > 0) || !(rp * 0.0 == 0.0)) my_unc = 1;
                d4 y;
#pragma unroll
                for (int r = 0; r < 3; ++r) y[r] = fma(fma(theta, nsig[r], ub[r]), (4 * r + g < pb.n && j < pb.n) ? 1.0 : 0.0, -sb[r]);      // S_b^-1 + Ubar - theta Sigbar
                y[3] = 0.0;
                // w = S_b^-1 s_b by column, then by component on the lanes of column 12
                const double wpart = -((sb[0] * s0 + sb[1] * s1) + sb[2] * s2);
                const double wcol = rows_sum4(wpart);
                wpad[j] = wcol;                                                    // (every row writes the same value)
                pd = 1; rp = 1.0;
                elim_round<0, 0>(y, em, pd, rp);
                elim_round<1, 0>(y, em, pd, rp);
                elim_round<2, 0>(y, em, pd, rp);
                elim_round<3, 0>(y, em, pd, rp);
                elim_round<4, 0>(y, em, pd, rp);
                elim_round<5, 0>(y, em, pd, rp);                                   // y = -Xt
                if (!(pd > 0) || !(rp * 0.0 == 0.0)) my_unc = 1;
                WAVE_SYNC();
                d4 aa, aan;
#pragma unroll
                for (int r = 0; r < 3; ++r) { aa[r] = fma(wpad[4 * r + g], mcol12, ac[r]); aan[r] = -aa[r]; }
                aa[3] = 0.0; aan[3] = 0.0;
                const d4 tgn = mm3(y, aa, zero4);                                  // -Xt Aa
                vt = mm3(aan, tgn, v);                                             // Jv + Aa' Xt Aa   (v = Jv after phase 1)
                vt[3] *= mA;                                                       // row 12: s_vec'; the additive scalar [12][12] is summed over the
                                                                                   // segments at the end, not propagated
            }
            if (wave >= 1) post(vt, wave - 1, epoch);                              // (always: nobody downstream may hang)
            PSW_MARK(3);
            // ---- phase 3: the ordinary recursion over segment `wave` from the value the hop produced -----------------------------------------
            v = vt;
            thi = pc.cut[wave + 1]; tlo = pc.cut[wave]; tmid = tlo;
        }
        // ---- the ordinary recursion: the last wave over its two segments (posting at the cut between them), the others over their own --------
        racc = 0.0; rprod = 1.0; rexp = 0;
        if (!my_unc && !dead) {
            int r = 0;
            for (int part = 0; part < 2 && !r; ++part) {                           // (one call site: one copy of the loop)
                const int hi = part ? tmid : thi, lo = part ? tlo : tmid;
                if (hi > lo) r = run(FalseTag(), hi, lo);
                if (part == 0 && wave == P - 1) {
                    if (r && l == 0) __hip_atomic_store(&sh->last_rc, (epoch << 2) | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    PSW_MARK(1);
                    post(v, P - 2, epoch);                                         // (always: whatever it holds ends the others' wait)
                    PSW_MARK(3);
                }
            }
            if (r && wave == P - 1 && l == 0) __hip_atomic_store(&sh->last_rc, (epoch << 2) | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (r == 1) my_fail = 1;
            if (r == 2) my_h = 1;
        }
        PSW_MARK(4);
        // ---- per-wave partials; one decision for the workgroup ----------------------------------------------------------------------------
        {
            const double rsum = wave_sum(racc);
            const double lterm = log(rprod) + (double)rexp * 0.6931471805599453094;
            if (l == 12) { sh->part[wave][0] = 0.5 * v[3]; sh->part[wave][1] = rsum; sh->part[wave][2] = lterm; }
            if (l == 0) {
                if (my_fail) __hip_atomic_store(&sh->fail_def, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (my_unc) __hip_atomic_store(&sh->uncertain, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (my_h) __hip_atomic_store(&sh->hnotpd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        team_barrier();
        PSW_MARK(5);
        const int any_fail = wave_uniform(sh->fail_def), any_h = wave_uniform(sh->hnotpd), v_last = wave_uniform(sh->last_rc);
        const bool last_ended = (v_last >> 2) == epoch;
        const int any_unc = last_ended ? 0 : wave_uniform(sh->uncertain);       // (what ended the last segment is definite whatever the others saw)
        if (any_unc) {                                   // the element form could not decide: the sequential sweep, on wave 0
            team_barrier();                              // (everyone has read the words of this attempt)
            if (wave == 0) sweep_body<GAIN, false, WM, HASL, 0, FLY>(a, tid, wls);
            return;
        }
        fail = any_fail ? 1 : 0;
        if (GAIN && !fail && any_h) {
            // increase_mu_and_delta!  (ileqg.jl:471-474), then the whole sweep again (:373-378); every wave keeps the same mu, delta
            delta = fmax(a.op.delta_0, delta * a.op.delta_0);
            mu = fmax(a.op.mu_min, mu * delta);
            if (++restarts > 400 || !isfinite(mu)) { fail = 5; break; }
            team_barrier();                              // (everyone has read the words of this attempt before wave 0 clears them)
            continue;
        }
        break;
    }
    if (wave != 0) return;
    double s_half = 0.0, s_racc = 0.0, s_log = 0.0;
    for (int w = P - 1; w >= 0; --w) { s_half += sh->part[w][0]; s_racc += sh->part[w][1]; s_log += sh->part[w][2]; }     // in the recursion's order
    const double tot = 0.5 * s_racc + coef * s_log;
    if (l == 12) {
        const double s0 = s_half + tot;
        if (a.mode == 1) {
            st.value_c[cidx] = s0;
            st.flag_c[cidx] = fail ? 1 : 0;
        } else if (a.mode == 2) {
            st.value[b] = fail ? INFINITY : s0;
            if (fail) st.status[b] = 1;
        } else if (a.mode >= 4) {
            st.mu_spec[b] = mu;
            st.delta_spec[b] = delta;
            st.spec_st[b] = fail ? (fail == 1 ? 2 : 5) : 1;
        } else {
            st.mu[b] = mu;
            st.delta[b] = delta;
            st.iter[b] = xld(&st.iter[b]) + 1;
            if (fail) { st.status[b] = (fail == 1) ? 2 : 5; st.value[b] = INFINITY; }
            else {
                st.ls_eps[b] = xld(&st.eps_init[b]);
                st.ls_count[b] = 0;
                st.ls_active[b] = 1;
            }
            if (a.op_out) { a.op_out[0] = s0; a.op_out[1] = (double)(fail ? (fail == 1 ? 2 : 5) : 0); }
        }
    }
#undef HBUF
#undef FBUF
#undef SVB
}
