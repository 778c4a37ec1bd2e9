// kernels.hip -- gfx950 (CDNA4) kernels of the iLEQG hot path.
//
//   solve_fused_kernel  the COMPLETE solve! of one theta-sample in one persistent wavefront (default, E = 1): the phases below as
//                       device functions (rollin_body, sweep_body, sweep_dual_body, ls_select_body)          (ileqg.jl:635-659)
//   rollin_kernel       simulate_dynamics + approximate_model fused: rollout on the matrix pipe, tile records streamed to HBM
//                                                                                                (ileqg.jl:18-38, 62-87, 258-322)
//   sweep_kernel        risk-sensitive Riccati sweep, gain / policy evaluation                   (ileqg.jl:341-406 / 412-465)
//   rollout_kernel, linearize_kernel   the same two reference functions unfused (operator entry points)
//   noisy_rollout_kernel               simulate_dynamics with process noise + integrate_cost     (ileqg.jl:44-55, 94-109, 115-124)
//   ls_select_kernel / init_state_kernel / commit_init_kernel   per-sample control flow of step! / line_search! / solve!
//                                                                          (ileqg.jl:494-592, 598-613, 635-659) replayed on device
//   pets_rollout_kernel, pets_mean_kernel   PETS stochastic rollouts                             (pets.jl:76-157)
//
// Sweep: ONE WAVEFRONT PER TRAJECTORY.  The value function is carried as the augmented
// symmetric matrix V = [[S, s_vec], [s_vec', 2 s]] (13 x 13 inside a 16 x 16 tile) held in the
// accumulator layout of v_mfma_f64_16x16x4_f64 (4 doubles per lane: col = lane & 15,
// row = 4*reg + (lane >> 4)).  In that layout register s of a matrix X is at once
//   - the B operand of K-slice s of  (.) * X, and
//   - the A operand of K-slice s of  X' * (.)
// so every product of one backward step chains through registers with no data movement:
//     X  = V [A|B]            3 MFMA      (independent of the inverse)
//     M^-1                    6 MFMA      (M = W^-1 - theta S: symmetric sweep with 2x2 block pivots, one rank-2 MFMA per round;
//                                          elim_round in device_utils.h)
//     Y  = theta M^-1 X       3 MFMA
//     T  = X + V Y            3 MFMA      (= (D S)[A|B]; row 12 of T is (D s_vec)'[A|B])
//     F  = [A|B]' T + C       3 MFMA      (C = [[Q,P'],[P,R]] enters as the accumulator input)
//     V  = Fx + La' Ua + Ga' La   2 MFMA  (La = [L|dl], Ga = [G|g], Ua = H La + Ga: 4 x 16 "natural" rows)
// The 4x4 system H X = -[G|g] is solved redundantly by every lane for its own column (LDL').
// logdet(W M) is accumulated as a wave-uniform normalised running product of det(P_k) / (e_k e_k+1) (pivot blocks of M over
// pivots of W^-1) and reduced once per sweep.  See tests/step_model.py for the NumPy model of this step.
#include <hip/hip_runtime.h>
#include <math.h>

#include "layout.h"
#include "kernels.h"

#include "device_utils.h"
#include "rat_normal.h"
#include "rat_pow.h"
#include "sweep_dual.h"

// Translation-unit parts: every part sees every device body (templates, inlined where instantiated); the __global__ kernels and their
// launchers are compiled by exactly one part.  No RAT_PART: the whole file in one unit.
#define PART_SWEEP 1      /* sweep_kernel and its launchers */
#define PART_ROLL 2       /* rollout / rollin / linearise kernels, per-sample control-flow kernels, small launchers */
#define PART_FUSED 4      /* solve_fused_kernel */
#define PART_BLOCK2 8     /* solve_block_kernel, E = 1 (two waves, padded geometry) */
#define PART_BLOCK3 16    /* E = 2 */
#define PART_BLOCK5 32    /* E = 4 */
#define PART_BLOCK8 64    /* E = 8 */
#define PART_MISC 128     /* PETS and noisy Monte-Carlo rollouts */
#define PART_PSW 256      /* psweep_kernel: the segment-parallel sweep (psweep.h) */
#define PART_BPSW 512     /* solve_block_psw_kernel: the workgroup-per-sample solve with segment-parallel sweeps */
#define PART_BPSW1 1024   /* ... its time-varying-W(k) instantiations (built without -amdgpu-mfma-vgpr-form: see launch_solve_block_psw_tv) */
#ifndef RAT_PART
#define RAT_PART 2047
#endif

#ifndef OCC2_PREFETCH
#define OCC2_PREFETCH 5            /* the same for the two-samples-per-SIMD solve: the other wave covers the latency, registers are scarce */
#endif
#ifndef OCC2_SWZ
#define OCC2_SWZ 1                 /* ... and the elimination's row exchange through the LDS crossbar (fewer vector instructions) */
#endif
#ifndef ROLLIN_PREFETCH
#define ROLLIN_PREFETCH 5          /* rotating operand sets of rollin_body: prefetch distance ROLLIN_PREFETCH - 1 steps */
#endif

// =====================================================================================================
// sweep_kernel
// =====================================================================================================
struct TileRegs {
    d4 z, c;
    double x, la;          // x: lanes 0..15 = qr, lane 16 = q;  la: this lane's entry L[g][j] of the natural 4 x 16 gain layout
    double xj;             // FLY: x_t[j] of this lane's column (the lanes holding a diagonal element of f_x need it)
};

// All loads are unconditional and branch-free (clamped lane offsets): the number of loads in flight is then a
// compile-time constant, so the prefetch of step t-1 can stay outstanding across the whole of step t behind a counted
// s_waitcnt vmcnt(N).  (A lane-conditional load makes the count path-dependent and degrades the wait to ~vmcnt(0),
// which exposes a full HBM latency per time step.)  The record is a register image (layout.h): three 16-B/lane loads
// fetch R0..R5, two 8-B/lane loads R6 and the [qr | q] row; policy evaluation adds one for its gain row.
template <bool HASL, bool DUMP, int FLY = 0>
__device__ __forceinline__ void load_tile(TileRegs &tr, const double *__restrict__ tp, int l, int lx, int lq,
                                          const double *__restrict__ Lp, const double *__restrict__ dlp, double mL, int g, int j,
                                          const FlyCtx *fc = nullptr, int t = 0) {
    if (FLY) {
        tr.xj = xld(&fc->xh[(long)t * XSTR + ((j < 12) ? j : 11)]);
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
    tr.x = xld(&tp[TS_QR + lx]);
    tr.la = 0.0;
    if (HASL) {
        tr.la = xld(&Lp[lq]) * mL;             // mL = 1 on lanes that hold a gain column (j < 12), else 0
        if (DUMP && dlp && j == 12) tr.la = dlp[g];      // operator form only (rat_dp_policy_eval with a dl_array)
    }
}

// HASL (policy evaluation only): a gain history is given.  false for initialize!'s sweep (ileqg.jl:221-224), whose gains
// are all zero: no gain loads and V = Fx without the two rank-4 updates.
// sweep_body is the whole sweep of ONE wavefront (trajectory `tid` of the launch); sweep_kernel wraps it one block per
// trajectory, solve_fused_kernel calls it as one phase of a sample's complete solve.
// wls: this wavefront's LDS scratch (WLS_SWEEP doubles).
// SWZ: the elimination's row exchange through the LDS crossbar (ds_swizzle) instead of vector-ALU lane swaps: identical values, fewer
// vector instructions, longer latency -- for the kernel that runs two samples per SIMD, which is short of issue slots, not of latency.
template <bool GAIN, bool DUMP, int WM, bool HASL, int SWZ = 0, int FLY = 0, bool PRUNE = false>
__device__ __forceinline__ void sweep_body(const SweepArgs &a, const int tid, double *const wls) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));      // opaque per phase: keeps the per-lane constants of one phase from being shared with
                                         // (and kept live across) the other phases inlined into solve_fused_kernel
    const int l_ = lane_, g_ = l_ >> 4, j_ = l_ & 15;
    const int l = l_, g = g_, j = j_;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    [[maybe_unused]] constexpr int dgs = (GAIN ? 0 : (HASL ? 4 : 8));
    BODY_MARK(a.dump, dgs + 0);
    // (every per-sample scalar is fetched before the first test on any of them: issued back to back the loads are in flight
    //  together; tested one by one they are a chain of dependent L2 round trips at every phase boundary of the fused solve)
    int b, k = 0, slot, cidx = -1;
    if (a.mode == 1) { const int Ek = st.E - a.k_first; b = tid / Ek; k = a.k_first + (tid - b * Ek); } else b = tid;
    const int fidx = b * st.E + k;
    const int v_act = xld(&st.ls_active[b]), v_flag = xld(&st.flag_c[fidx]), v_nom = xld(&st.slot_nom[b]), v_stat = xld(&st.status[b]), v_sel = xld(&st.lsel[b]);
    const int s_act = wave_uniform(v_act), s_flag = wave_uniform(v_flag), s_nom = wave_uniform(v_nom), s_stat = wave_uniform(v_stat), sel = wave_uniform(v_sel);
    const double theta = xld(&st.theta[b]), mu_in = xld(&st.mu[b]);
    double delta = xld(&st.delta[b]);
    if (a.mode == 1) {
        if (!s_act) return;
        cidx = fidx;
        if (s_flag == 2) return;
        // PRUNE (candidates 1 .. E-1 of the round-based path): candidate 0 is already known to be the line search's choice -- the sequential
        // rule will not read this candidate's value (ls_select_body stops at the first it accepts)
        if (PRUNE && __builtin_amdgcn_readfirstlane(__atomic_load_n(&st.acc0[b], __ATOMIC_RELAXED))) return;
        slot = cand_slot(b, k, s_nom, st.E);
    } else if (a.mode == 4) {          // speculative gain sweep of the NEXT iteration on line-search candidate 0
        if (!s_act) return;
        if (s_flag == 2) return;
        slot = cand_slot(b, 0, s_nom, st.E);
    } else if (a.mode == 5) {          // speculative first gain sweep on the nominal tiles, concurrent with initialize!'s sweep
        slot = b * (st.E + 1) + s_nom;
    } else {
        if (s_stat != ST_RUNNING) return;
        if (a.mode == 0 && s_act) return;                 // still inside line_search! of its current iteration
        slot = b * (st.E + 1) + s_nom;
    }
    double mu = (a.mode == 2) ? 0.0 : (a.mode == 3 ? a.mu_op : mu_in);
    // a speculative gain sweep running as its own wavefront shares its SIMD with evaluation waves of the same round and is the longer
    // chain: it takes issue priority (the evaluation wave beside it fills the slots this wave's dependency stalls leave)
    if (GAIN && a.mode >= 4) __builtin_amdgcn_s_setprio(3);
    const int N = st.N;
    const double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot) * st.tile_stride;
    FlyCtx fc;
    if (FLY) fly_init(fc, pb, st.xs + (long)slot * st.x_stride, l, g, j, FLY == 2);
    const int osel = (a.mode >= 4) ? (sel ^ 1) : sel;      // speculative sweeps fill the other half; committed by select
    const double *__restrict__ Lb = st.L + (long)sel * st.l_half + (long)b * N * LSTR;
    double *__restrict__ Lout = st.L + (long)osel * st.l_half + (long)b * N * LSTR;
    double *__restrict__ dlout = st.dl + (long)osel * st.dl_half + (long)b * N * USTR;

    double *const lbuf = wls;          // policy evaluation: the step's gain row block [L | dl], natural 4 x 16 layout
    double *const ex = wls + 64;       // exchange area: rows 0..3 = [G | H] (4 x 16), row 4 = f (16), [80] = 0.0, [84..99] = s_vec,
                                       // [104..167] dump slots: the row-0-only writes are unconditional, idle lanes write there
                                       // (a lane-conditional write splits the basic block the scheduler works on)
#define HBUF(r_, c_) ex[(r_) * 16 + (c_)]
#define FBUF(c_) ex[64 + (c_)]
#define SVB(c_) ex[84 + (c_)]
    if (l < 8) ex[80 + l] = 0.0;

    // The loop is bound by VALU issue of ONE wave (every instruction costs >= 4 cycles), so lane-position selects are
    // replaced by per-lane 0/1 multipliers and per-lane LDS offsets computed once here (loop invariant, kept in VGPRs).
    const double mL = ((a.mode == 1 || a.mode == 3) && j < 12) ? 1.0 : 0.0;
    double m12 = (j < 12) ? 1.0 : 0.0;                       // lanes holding a state column
    double nth12 = -st.theta[b] * m12;
    double mA = (g == 0 && j < 12) ? 1.0 : 0.0, mB = (g == 0 && j == 12) ? 1.0 : 0.0, mH = (j == 12 + g) ? 1.0 : 0.0;
    ElimMasks em;
    elim_masks(em, g, j);
    int hoff[4], foff[3], goff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hoff[k] = (g <= k) ? (g * 16 + 12 + k) : (k * 16 + 12 + g);      // Symmetric(H): upper triangle (:371)
        goff[k] = (j < 12) ? (k * 16 + j) : (j == 12 ? 64 + 12 + k : 80); // column j of [G | g | 0]
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) foff[r] = (j == 12) ? (64 + 4 * r + g) : 80;
    const int gaoff = (j == 12) ? (64 + 12 + g) : 80;

    const int svo = (g == 0) ? 84 + j : 104 + l, fbo = (g == 0) ? 64 + j : 104 + l;
    double *const pgl = (j < 12) ? Lout + g * 12 + j : (j == 12 ? dlout + g : sample_sink(st, b) + l);
    const long sgl = (j < 12) ? LSTR : (j == 12 ? USTR : 0);
    const int lx = (l < 17) ? l : TS_PAD - TS_QR;            // [qr | q] row: lanes past q read the record's zero slot
    const int lq = g * 12 + ((j < 12) ? j : 11);             // own entry of L_t (4 x 12 row-major), clamped

    // noise tables (time-invariant case is hoisted out of the time loop)
    d4 winv = {0, 0, 0, 0}, wp = {0, 0, 0, 0};
    double epall = 1.0;                     // prod over the six pivot blocks of 1/(e_k e_k+1), e = pivots of inv(W) (wave-uniform)
    // WM == 2 (W diagonal, time-invariant): (D S)[A|B] is formed as V M^-1 inv(W) [A|B] -- D S = S M^-1 inv(W) exactly, since
    // inv(W) = M + theta S -- with inv(W) folded into the rows of M^-1's A operand: 6 MFMAs where X = V [A|B], theta M^-1 X and
    // X + V (theta M^-1 X) take 9.  nwrow[r] = -inv(W)_ii of this lane's row i = 4 r + g (the sweep leaves -M^-1).
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
    const double coef = (theta != 0.0) ? -1.0 / (2.0 * theta) : 0.0;      // of logdet(W M); traces / quadratic forms carry 1/2

    DIAG_DECL
    int restarts = 0;
    int fail = 0;         // 1: M not PD, 5: mu diverged
    d4 v;
    double racc, rprod;
    int rexp;
    while (true) {        // mu-regularisation restart loop (ileqg.jl:359); runs once for policy evaluation
        // terminal condition (ileqg.jl:352-354 / 429-431)
        const double *__restrict__ tt = tile0 + (long)N * TSTRIDE;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = 4 * r + g;
            // (one unconditional load per register, the address selected per lane: conditional loads compile to a chain of divergent
            //  branches, each waiting for its own round trip -- 4.3 k cycles of prologue measured in the fused kernel)
            const double te = xld(&tt[(j < 12) ? TT_Q + i * 12 + j : TT_QV + i]);
            v[r] = (j <= 12) ? te : 0.0;
        }
        const double t3 = xld(&tt[(j < 12) ? TT_QV + j : TT_q]);
        v[3] = (g == 0 && j <= 12) ? (j < 12 ? t3 : 2.0 * t3) : 0.0;
        racc = 0.0;
        rprod = 1.0;
        rexp = 0;
        if (DUMP) {
            double *dp = a.dump + (long)N * DUMP_STRIDE;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int i = 4 * r + g;
                if (j < 12) dp[DUMP_S + i * 12 + j] = v[r];
                if (j == 12) dp[DUMP_SV + i] = v[r];
            }
            if (l == 12) dp[DUMP_s] = 0.5 * v[3];
        }
        TileRegs nx;
        load_tile<HASL, DUMP, FLY>(nx, tile0 + (long)(N - 1) * TSTRIDE, l, lx, lq, Lb + (long)(N - 1) * LSTR,
                        a.dl_in ? a.dl_in + (long)(N - 1) * USTR : nullptr, mL, g, j, &fc, N - 1);
        bool h_not_pd = false;
        // one backward step (ileqg.jl:361-391 / :435-460) on the tile registers `cur`; returns 0, 1 (M not PD), 2 (H not PD)
        auto step = [&](const int t, const TileRegs &cur) -> int {
            // opaque per-step copies of the lane indices: keeps the (lane == const) masks as one v_cmp at their use
            // instead of loop-invariant SGPR pairs (which spill: the kernel is SGPR-bound, not VGPR-bound)
            int l = l_, g = g_, j = j_;
            asm volatile("" : "+v"(l), "+v"(g), "+v"(j));
            DIAG_START();
            if (WM == 1) {          // time-varying W(k): separate instantiation, so that the common case keeps a static load count
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    winv[r] = pb.Winv[(long)t * 192 + 64 * r + l];
                    wp[r] = pb.Wp[(long)t * 192 + 64 * r + l];
                }
                const double *ept = pb.epiv + (long)t * 16;
                epall = ((((ept[0] * ept[2]) * ept[4]) * ept[6]) * ept[8]) * ept[10];
            }
            // the step's tile: [A | B] and [[Q, 0], [P, R]] from the record, or (FLY) formed from x_t and the problem tables
            d4 cz, ccs;
            if (!FLY) { cz = cur.z; ccs = cur.c; }
            else {
#pragma unroll
                for (int r = 0; r < 3; ++r) cz[r] = fx_diag(fc.zt[r], fc.dg[r], fc.kappa, cur.xj);
                cz[3] = 0.0;
                if (FLY == 2) { ccs[0] = cur.c[0] * fc.mq; ccs[1] = cur.c[1] * fc.mq; ccs[2] = cur.c[2] * fc.mq; ccs[3] = cur.c[3]; }
                else ccs = fc.cc;
            }
            // X = V[:, 0:12] [A|B] (rows 0..11 = S [A|B], row 12 = s_vec'[A|B]).  Issued first: it does not depend on the
            // inverse, so the matrix pipe works through it while the VALU runs the elimination below.  (Diagonal W: only theta == 0 needs it.)
            // (WM == 2 forms it inside the theta == 0 branch: a conditionally defined xz costs a zero fill of eight registers per step)
            d4 xz;
            if (WM != 2) xz = mm3(v, cz, (d4){0, 0, 0, 0});
            if (HASL) lbuf[l] = cur.la;                                     // rows of [L | dl] to every lane (read after the next fence)
            const double qc = readlane_f64(cur.x, 16);                      // c (ileqg.jl:296)
            d4 tm;
            if (theta != 0.0) {
                ex[svo] = v[3];                                             // s_vec (row 12 of V) to every lane, off the critical path
                // M = Symmetric(inv(W) - theta S)   (ileqg.jl:365)
                d4 m;
#pragma unroll
                for (int r = 0; r < 3; ++r) m[r] = fma(nth12, v[r], winv[r]);     // inv(W) table is zero in columns >= 12
                m[3] = 0.0;
                DIAG_STAMP(0, m[0]);
                // symmetric sweep operator with 2x2 block pivots K = {k, k+1}, k = 0, 2, .., 10:  m <- -M^-1.
                // With P = M_KK, Bk = P^-1:  M'_KK = -Bk, M'_Kj = Bk M_Kj, M'_iK = M_iK Bk, M'_ij = M_ij - M_iK Bk M_Kj.
                // Rows k, k+1 (with -I in their pivot slots) are their own A operand, so that every entry obeys the one formula
                //   m'_ij = base_ij - vi1 U1_j - vi2 U2_j,   U_j = Bk [v1_j; v2_j],   base = 0 on pivot rows/columns, m elsewhere.
                // Leading minors p11 > 0, det P > 0 for every block  <=>  isposdef(M)  (:366); det P = d_k d_{k+1}.
                int pdmin = 1;                                               // min over the high words of the leading minors (elim_round)
                rprod *= epall;
                elim_round<0, SWZ>(m, em, pdmin, rprod);
                elim_round<1, SWZ>(m, em, pdmin, rprod);
                elim_round<2, SWZ>(m, em, pdmin, rprod);
                elim_round<3, SWZ>(m, em, pdmin, rprod);
                elim_round<4, SWZ>(m, em, pdmin, rprod);
                elim_round<5, SWZ>(m, em, pdmin, rprod);
                DIAG_STAMP(1, m[0]);
                // (a NaN or infinite pivot block poisons the running product of the determinants: the tripwire beside the sign test)
                if (!(pdmin > 0) || !(rprod * 0.0 == 0.0)) { fail = 1; return 1; }
                // (the product is renormalised once per step: one log() per sweep instead of twelve per step)
                rexp += __builtin_amdgcn_frexp_exp(rprod);
                rprod = __builtin_amdgcn_frexp_mant(rprod);
                if (WM == 2) {
                    // theta s_vec' M^-1 s_vec (:387), per lane: (-theta s_j) sum_i (-M^-1)_ij s_i (columns >= 12 of the sweep's result are zero)
                    racc += (nth12 * SVB(j)) * (m[0] * SVB(g) + m[1] * SVB(4 + g) + m[2] * SVB(8 + g));
                    // T = V M^-1 inv(W) [A|B]: rows 0..11 = (D S)[A|B], row 12 = (D s_vec)'[A|B]   (:367)
                    d4 mw;
#pragma unroll
                    for (int r = 0; r < 3; ++r) mw[r] = m[r] * nwrow[r];
                    mw[3] = 0.0;
                    const d4 y2 = mm3(mw, cz, (d4){0, 0, 0, 0});
                    tm = mm3(v, y2, (d4){0, 0, 0, 0});
                } else {
                    // theta M^-1 (the sweep left -M^-1); padded columns cleared
                    d4 minv;
#pragma unroll
                    for (int r = 0; r < 3; ++r) minv[r] = nth12 * m[r];
                    minv[3] = 0.0;
                    // theta s_vec' M^-1 s_vec (:387): the constant term never feeds back into S, s_vec or the gains, so it is
                    // accumulated per lane (sum_ij (theta M^-1)_ij s_i s_j) and reduced once per sweep with the other scalars
                    racc += SVB(j) * (minv[0] * SVB(g) + minv[1] * SVB(4 + g) + minv[2] * SVB(8 + g));
                    // T = (D S)[A|B] = X + V (theta M^-1 X): rows 0..11 = (D S)[A|B], row 12 = (D s_vec)'[A|B]   (:367)
                    const d4 y2 = mm3(minv, xz, (d4){0, 0, 0, 0});
                    tm = mm3(v, y2, xz);
                }
                DIAG_STAMP(2, tm[0]);
            } else {
                // theta == 0: D = I ; 0.5 tr(W S)   (ileqg.jl:385).  The reference still forms M = inv(W) - 0 S and asserts
                // isposdef(M) (:365-366 / :439-440): a non-finite entry of S makes M NaN, i.e. not PD (0 x Inf = NaN).
                const double nf = fma(v[2], 0.0, fma(v[1], 0.0, v[0] * 0.0));
                if (__ballot(nf != nf) & 0x0FFF0FFF0FFF0FFFull) { fail = 1; return 1; }      // lanes j < 12 hold S
                // (explicit fma: the same contraction in every instantiation and in sweep_dual_body -- the paths are tested bit for bit)
                racc = fma(m12, fma(wp[2], v[2], fma(wp[1], v[1], wp[0] * v[0])), racc);
                if (WM == 2) tm = mm3(v, cz, (d4){0, 0, 0, 0}); else tm = xz;
            }
            // F = [A|B]' T + [[Q,P'],[P,R]]  (:369-370 and the Q + A'DSA term of :390)
            d4 f = mm3(cz, tm, ccs);
            // H block: rows 12..15 of F live in register 3;  + mu I  (:370)
            const double gh = fma(mu, mH, f[3]);
            const double fv = tm[3] + cur.x;      // lanes g == 0: [q_vec + A' D s_vec | r + B' D s_vec]  (:368, :389)
            DIAG_STAMP(3, gh);
            HBUF(g, j) = gh;
            ex[fbo] = fv;
            WAVE_SYNC();
            // row g of H = Symmetric(H) (upper triangle, :371) and this lane's entry of [G | g | 0]
            const double hg0 = ex[hoff[0]], hg1 = ex[hoff[1]], hg2 = ex[hoff[2]], hg3 = ex[hoff[3]];
            const double ga = fma(gh, m12, ex[gaoff]);
            double x0, x1, x2, x3, la;
            if (GAIN) {
                const double h00 = HBUF(0, 12), h01 = HBUF(0, 13), h02 = HBUF(0, 14), h03 = HBUF(0, 15);
                const double h11 = HBUF(1, 13), h12 = HBUF(1, 14), h13 = HBUF(1, 15);
                const double h22 = HBUF(2, 14), h23 = HBUF(2, 15), h33 = HBUF(3, 15);
                const double g0 = ex[goff[0]], g1 = ex[goff[1]], g2 = ex[goff[2]], g3 = ex[goff[3]];   // column j of [G | g | 0]
                // LDL' of H; all pivots > 0 <=> isposdef(H)   (:372)
                const double d0 = h00, i0 = fast_rcp(d0);
                const double l10 = h01 * i0, l20 = h02 * i0, l30 = h03 * i0;
                const double d1 = h11 - l10 * h01, i1 = fast_rcp(d1);
                const double l21 = (h12 - l20 * h01) * i1, l31 = (h13 - l30 * h01) * i1;
                const double d2 = h22 - l20 * h02 - l21 * (l21 * d1), i2 = fast_rcp(d2);
                const double l32 = (h23 - l30 * h02 - l31 * (l21 * d1)) * i2;
                const double d3 = h33 - l30 * h03 - l31 * (l31 * d1) - l32 * (l32 * d2), i3 = fast_rcp(d3);
                if (!(d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) { h_not_pd = true; return 2; }
                // X = -H \ [G | g]   (:379-382)
                const double y0 = -g0;
                const double y1 = -g1 - l10 * y0;
                const double y2 = -g2 - l20 * y0 - l21 * y1;
                const double y3 = -g3 - l30 * y0 - l31 * y1 - l32 * y2;
                x3 = y3 * i3;
                x2 = y2 * i2 - l32 * x3;
                x1 = y1 * i1 - l21 * x2 - l31 * x3;
                x0 = y0 * i0 - l10 * x1 - l20 * x2 - l30 * x3;
                // [L | dl] natural rows: row g takes x_g (0/1 row selectors, not a select between computed values: that compiles
                // to a chain of divergent branches)
                la = ((x0 * em.e0[0] + x1 * em.e1[0]) + x2 * em.e0[1]) + x3 * em.e1[1];
            } else if (HASL) {
                x0 = lbuf[j]; x1 = lbuf[16 + j]; x2 = lbuf[32 + j]; x3 = lbuf[48 + j];   // column j of [L | dl]
                la = cur.la;
            } else {
                x0 = x1 = x2 = x3 = la = 0.0;
            }
            const double ua = hg0 * x0 + hg1 * x1 + hg2 * x2 + hg3 * x3 + ga;         // H [L|dl] + [G|g]
            DIAG_STAMP(4, ua);
            if (GAIN) {
                pgl[(long)t * sgl] = (j <= 12) ? la : 0.0;      // L_t (columns 0..11) | dl_t (column 12) | idle lanes: sink
            }
            // Fx = [[Q + A'DSA, f_x], [f_x', 2q + 2s + theta s'M^-1 s]]
            d4 fx;
#pragma unroll
            for (int r = 0; r < 3; ++r) fx[r] = fma(f[r], m12, ex[foff[r]]);
            fx[3] = fma(fv, mA, (2.0 * qc + v[3]) * mB);
            // V = Fx + La' Ua + Ga' La    (:383, :389, :390)
            if (GAIN || HASL) {
                d4 vn = MFMA(la, ua, fx);
                vn = MFMA(ga, la, vn);
                v = vn;
            } else {
                v = fx;
            }
            DIAG_STAMP(5, v[0]);
            if (DUMP) {
                double *dp = a.dump + (long)t * DUMP_STRIDE;
                const double tot = sweep_scalars(wave_sum(racc), coef, rprod, rexp, theta != 0.0);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = 4 * r + g;
                    if (j < 12) dp[DUMP_S + i * 12 + j] = v[r];
                    if (j == 12) dp[DUMP_SV + i] = v[r];
                }
                if (l == 12) dp[DUMP_s] = 0.5 * v[3] + tot;
                if (l < 4) dp[DUMP_g + l] = FBUF(12 + l);
                if (j < 12) dp[DUMP_G + g * 12 + j] = HBUF(g, j);
                else dp[DUMP_H + g * 4 + (j - 12)] = (g <= j - 12) ? HBUF(g, j) : HBUF(j - 12, 12 + g);
            }
            WAVE_SYNC();          // the exchange area is rewritten next step
            return 0;
        };
        // Time loop, unrolled by two over ping-pong tile registers: the software prefetch of step t-1 (a full step of
        // latency hiding) lands directly in the registers the next step reads, with no copies.  The prefetch is
        // unconditional (index clamped at 0) so that the loads stay in straight-line code and the compiler can wait
        // with a counted vmcnt(N) instead of vmcnt(0) at the join of a branch.
        TileRegs rb;
        BODY_MARK(a.dump, dgs + 1);
        for (int t = N - 1; t >= 0; t -= 2) {
            // (PRUNE: the word is requested ahead of the pair's tile loads and looked at behind its two steps -- the oldest load in flight by
            //  then, so the counted wait leaves every younger one alone)
            int polled = 0;
            if (PRUNE) polled = __atomic_load_n(&st.acc0[b], __ATOMIC_RELAXED);
            {
                const int tn = (t > 0) ? t - 1 : 0;
                load_tile<HASL, DUMP, FLY>(rb, tile0 + (long)tn * TSTRIDE, l, lx, lq, Lb + (long)tn * LSTR,
                                a.dl_in ? a.dl_in + (long)tn * USTR : nullptr, mL, g, j, &fc, tn);
            }
            if (step(t, nx)) break;
            if (t == 0) break;
            {
                const int tn = (t > 1) ? t - 2 : 0;
                load_tile<HASL, DUMP, FLY>(nx, tile0 + (long)tn * TSTRIDE, l, lx, lq, Lb + (long)tn * LSTR,
                                a.dl_in ? a.dl_in + (long)tn * USTR : nullptr, mL, g, j, &fc, tn);
            }
            if (step(t - 1, rb)) break;
            if (PRUNE && __builtin_amdgcn_readfirstlane(polled)) return;        // nobody will read this candidate: no outputs
        }
        BODY_MARK(a.dump, dgs + 2);
        if (GAIN && h_not_pd) {
            // increase_mu_and_delta!  (ileqg.jl:471-474), then restart the whole sweep (:373-378)
            delta = fmax(a.op.delta_0, delta * a.op.delta_0);
            mu = fmax(a.op.mu_min, mu * delta);
            if (++restarts > 400 || !isfinite(mu)) { fail = 5; break; }
            WAVE_SYNC();
            continue;
        }
        break;
    }
#ifdef RAT_DIAG
    if (l_ == 0 && blockIdx.x < 8 && a.dump)
        for (int q = 0; q < 6; ++q) a.dump[blockIdx.x * 8 + q] = (double)dg_acc[q];
#endif
    const double tot = sweep_scalars(wave_sum(racc), coef, rprod, rexp, theta != 0.0);
    if (l == 12) {
        const double s0 = 0.5 * v[3] + tot;
        if (a.mode == 1) {
            st.value_c[cidx] = s0;
            st.flag_c[cidx] = fail ? 1 : 0;
            if (a.prune && k == 0) {         // (workgroup-per-sample kernel, E > 1: see sweep_dual_body, mode 7)
                const double cur = st.value[b], eps = st.ls_eps[b];
                const bool take = !fail && (isapprox_default(s0, cur) || s0 < cur || eps * a.op.lambda < a.op.eps_min);
                __atomic_store_n(&st.acc0[b], take ? 1 : 0, __ATOMIC_RELAXED);
            }
        } else if (a.mode == 2) {
            st.value[b] = fail ? INFINITY : s0;
            if (fail) st.status[b] = 1;                      // RAT_ST_M_NOT_PD_INIT
        } else if (a.mode == 3) {
            a.op_out[0] = s0;
            a.op_out[1] = (double)(fail ? 2 : 0);
        } else if (a.mode >= 4) {
            st.mu_spec[b] = mu;
            st.delta_spec[b] = delta;
            st.spec_st[b] = fail ? (fail == 1 ? 2 : 5) : 1;
        } else {
            st.mu[b] = mu;
            st.delta[b] = delta;
            st.iter[b] = xld(&st.iter[b]) + 1;                                  // step!: iter_current += 1   (ileqg.jl:599)
            if (fail) { st.status[b] = (fail == 1) ? 2 : 5; st.value[b] = INFINITY; }   // M_NOT_PD_GAIN / MU_DIVERGED
            else {                                            // line_search! starts at eps_init   (ileqg.jl:502)
                st.ls_eps[b] = xld(&st.eps_init[b]);
                st.ls_count[b] = 0;
                st.ls_active[b] = 1;
            }
            if (a.op_out) { a.op_out[0] = s0; a.op_out[1] = (double)(fail ? (fail == 1 ? 2 : 5) : 0); }
        }
    }
    BODY_MARK(a.dump, dgs + 3);
#undef HBUF
#undef FBUF
#undef SVB
}

#include "psweep.h"

#if RAT_PART & PART_PSW
// One workgroup of pc.P wavefronts per trajectory (psweep.h).  The modes are sweep_kernel's (0 gain sweep, 1 policy evaluation of candidates,
// 2 initialize!'s evaluation, 4 / 5 speculative gain sweeps); results agree with sweep_kernel's to rounding (tests/test_gpu_psweep.py).
// MAXP = 4: one wavefront per SIMD, each may hold 512 registers (256 + 256 accumulation registers as spill space of the gain element's
// 4 x 4 factors).
template <bool GAIN, int WM, bool HASL, int FLY, int MAXP>
__global__ __launch_bounds__(64 * MAXP) void psweep_kernel(SweepArgs a, PswCuts pc) {
    __shared__ double wls[MAXP][WLS_PSW];
    __shared__ PswShared sh;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < PSW_MAXP) sh.flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) { sh.bar = 0; sh.last_rc = 0; sh.lastP = 0; }
    __syncthreads();
    psweep_body<GAIN, WM, HASL, FLY>(a, blockIdx.x, wls[wave], &sh, pc, wave);
}

bool psweep_supported(const SweepArgs &a, bool gain) {
    if (a.fly && a.pb.cost_tv) return false;
    if (gain) return a.mode == 0 || a.mode == 4 || a.mode == 5;
    return a.mode == 1 || a.mode == 2;
}

void launch_psweep(const SweepArgs &a, int ntraj, bool gain, const PswCuts &pc, hipStream_t s) {
    if (ntraj <= 0) return;
    if (pc.P > 4) return;                                    // (not reachable: psweep_cuts is asked for at most four waves)
    const dim3 grid(ntraj), block(64 * pc.P);
    const int wm = a.pb.W_tv ? 1 : (a.pb.W_diag ? 2 : 0);
    // (teams of up to four waves, one per SIMD: the eight-wave instantiations of rounds 4-5 -- two waves per SIMD, 256 registers each, gain
    //  elements spilling to scratch -- never won a measurement and were retired in round 6; the switch psweep is clamped to 4 on the host)
#define PSW_LAUNCH(G, H, F) do { if (wm == 2) hipLaunchKernelGGL((psweep_kernel<G, 2, H, F, 4>), grid, block, 0, s, a, pc); \
                                 else if (wm == 1) hipLaunchKernelGGL((psweep_kernel<G, 1, H, F, 4>), grid, block, 0, s, a, pc); \
                                 else hipLaunchKernelGGL((psweep_kernel<G, 0, H, F, 4>), grid, block, 0, s, a, pc); } while (0)
    if (gain) { if (a.fly) PSW_LAUNCH(true, false, 1); else PSW_LAUNCH(true, false, 0); }
    else if (a.mode == 2) { if (a.fly) PSW_LAUNCH(false, false, 1); else PSW_LAUNCH(false, false, 0); }
    else { if (a.fly) PSW_LAUNCH(false, true, 1); else PSW_LAUNCH(false, true, 0); }
#undef PSW_LAUNCH
}
#endif  // PART_PSW

#if RAT_PART & PART_SWEEP
// SWZ: the elimination's row exchange through the LDS crossbar (fewer vector instructions, longer latency; identical values): for launches
// that put several waves on a SIMD, where the datapath is saturated (profiles/r03_rocprof_summary.md: two evaluation waves per SIMD issue
// 50 % each) and only the instruction count matters
template <bool GAIN, bool DUMP, int WM, bool HASL, int FLY = 0, int SWZ = 0, bool PRUNE = false>
__global__ __launch_bounds__(64) void sweep_kernel(SweepArgs a) {
    __shared__ double wls[WLS_SWEEP];
    sweep_body<GAIN, DUMP, WM, HASL, SWZ, FLY, PRUNE>(a, blockIdx.x, wls);
}

template <bool GAIN, bool DUMP, bool HASL>
static void launch_sweep_w(const SweepArgs &a, dim3 grid, hipStream_t s) {
    if (a.pb.W_tv) hipLaunchKernelGGL((sweep_kernel<GAIN, DUMP, 1, HASL>), grid, dim3(64), 0, s, a);
    else if (a.pb.W_diag) hipLaunchKernelGGL((sweep_kernel<GAIN, DUMP, 2, HASL>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((sweep_kernel<GAIN, DUMP, 0, HASL>), grid, dim3(64), 0, s, a);
}

// Round-based path, candidate 0 of a line-search round (tile-free candidates): the paired pass (its evaluation + the NEXT step!'s gain sweep,
// sweep_dual_body mode 7) -- unless accepting the candidate would end solve! (d < d_tol with mu at its floor, or iter_max: ileqg.jl:642-653):
// nothing would consume the gains, and the plain evaluation (half the instructions) runs instead, as in solve_fused_kernel.  Same values either
// way (the recursions are independent); spec_st = 0 tells the accept rule that no gain sweep rides along.
template <int WM, int FLY>
__global__ __launch_bounds__(64) void sweep_cand0_kernel(SweepArgs a) {
    __shared__ double wls[WLS_DUAL];
    const StateDev &st = a.st;
    const int b = blockIdx.x;
    const int v_act = __atomic_load_n(&st.ls_active[b], __ATOMIC_RELAXED), v_it = __atomic_load_n(&st.iter[b], __ATOMIC_RELAXED);
    const double v_dc = *(const volatile double *)&st.d_c[b * st.E], v_mu = *(const volatile double *)&st.mu[b];
    if (!__builtin_amdgcn_readfirstlane(v_act)) return;
    const double dc = readlane_f64(v_dc, 0), mu = readlane_f64(v_mu, 0);
    const bool ends = (a.op.d > dc && mu <= a.op.mu_min) || __builtin_amdgcn_readfirstlane(v_it) == a.op.iter_max;
    if (ends) {
        if (threadIdx.x == 0) st.spec_st[b] = 0;
        SweepArgs sa = a; sa.mode = 1; sa.k_first = 0;
        sweep_body<false, false, WM, true, 1, FLY>(sa, b * st.E, wls);        // (tid = b E: candidate 0 of sample b when k_first = 0 and the launch covers E candidates)
    } else {
        sweep_dual_body<WM, true, FLY>(a, b, wls);
    }
}
void launch_sweep_cand0(const SweepArgs &a, int nsamples, hipStream_t s) {
    if (nsamples <= 0) return;
    const dim3 grid(nsamples), block(64);
#define C0_LAUNCH(W) do { if (a.pb.cost_tv) hipLaunchKernelGGL((sweep_cand0_kernel<W, 2>), grid, block, 0, s, a); \
                          else hipLaunchKernelGGL((sweep_cand0_kernel<W, 1>), grid, block, 0, s, a); } while (0)
    if (a.pb.W_tv) C0_LAUNCH(1); else if (a.pb.W_diag) C0_LAUNCH(2); else C0_LAUNCH(0);
#undef C0_LAUNCH
}

void launch_sweep(const SweepArgs &a, int ntraj, bool gain, bool dump, hipStream_t s) {
    if (ntraj <= 0) return;
    dim3 grid(ntraj);
    if (gain && a.fly && (a.mode == 4 || a.mode == 0) && !dump) {   // gain sweep on a trajectory whose record may hold only [c_x | c_u | c]: the
                                                                    // speculative one on candidate 0, and the plain one of a step! on whatever was accepted
#define FLYG_LAUNCH(W) do { if (a.pb.cost_tv) hipLaunchKernelGGL((sweep_kernel<true, false, W, false, 2>), grid, dim3(64), 0, s, a); \
                            else hipLaunchKernelGGL((sweep_kernel<true, false, W, false, 1>), grid, dim3(64), 0, s, a); } while (0)
        if (a.pb.W_tv) FLYG_LAUNCH(1); else if (a.pb.W_diag) FLYG_LAUNCH(2); else FLYG_LAUNCH(0);
#undef FLYG_LAUNCH
    } else if (gain) {
        if (dump) launch_sweep_w<true, true, false>(a, grid, s);
        else launch_sweep_w<true, false, false>(a, grid, s);
    } else if (a.mode == 2) {                      // initialize!: zero gains
        launch_sweep_w<false, false, false>(a, grid, s);
    } else if (a.fly && a.mode == 1 && !dump) {    // candidates whose records hold only [c_x | c_u | c]: tiles formed in the sweep
        const bool many = ntraj > 2048;            // more than two waves per SIMD on an MI355X: the datapath is saturated
#define FLY_LAUNCH(W) do { if (a.prune && a.pb.cost_tv) hipLaunchKernelGGL((sweep_kernel<false, false, W, true, 2, 0, true>), grid, dim3(64), 0, s, a); \
                           else if (a.prune) hipLaunchKernelGGL((sweep_kernel<false, false, W, true, 1, 1, true>), grid, dim3(64), 0, s, a); \
                           else if (a.pb.cost_tv) hipLaunchKernelGGL((sweep_kernel<false, false, W, true, 2>), grid, dim3(64), 0, s, a); \
                           else if (many) hipLaunchKernelGGL((sweep_kernel<false, false, W, true, 1, 1>), grid, dim3(64), 0, s, a); \
                           else hipLaunchKernelGGL((sweep_kernel<false, false, W, true, 1>), grid, dim3(64), 0, s, a); } while (0)
        if (a.pb.W_tv) FLY_LAUNCH(1); else if (a.pb.W_diag) FLY_LAUNCH(2); else FLY_LAUNCH(0);
#undef FLY_LAUNCH
    } else {
        if (dump) launch_sweep_w<false, true, true>(a, grid, s);
        else launch_sweep_w<false, false, true>(a, grid, s);
    }
}

#endif  // PART_SWEEP
// =====================================================================================================
// rollout_kernel: four trajectories per wavefront (one per 16-lane row).  Lane j < 12 owns state
// component j, lanes j < 4 additionally own control component j.
// =====================================================================================================
// x^e of the power-law family: the reference's own arithmetic (Julia -> openlibm's fdlibm pow, rat_pow.h), not the device library's pow --
// device and oracle agree bit for bit on every power, so long power-law iterations no longer drift apart
__device__ __forceinline__ double powchk(double bse, double e, int &dom) {
    const double r = rat_pow(bse, e);
    if (r != r && bse == bse) dom = 1;        // Julia: DomainError for a negative base with a fractional exponent
    return r;
}

#if RAT_PART & PART_ROLL
__global__ __launch_bounds__(64) void rollout_kernel(RolloutArgs a) {
    const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    const int ncand = (a.mode == 0) ? st.B : st.B * st.E;
    int c = blockIdx.x * 4 + row;
    bool live = c < ncand;
    int b = 0, k = 0;
    if (live) {
        if (a.mode == 0) { b = c; live = (st.status[b] == ST_RUNNING); }
        else { b = c / st.E; k = c - b * st.E; live = st.ls_active[b] != 0; }
    }
    if (__ballot(live) == 0ull) return;              // whole wave idle (finished samples, no-op rounds)
    __shared__ double shdx[4][12];
    __shared__ double shxu[4][16];
    __shared__ double shq[4][4];

    const int nom = live ? st.slot_nom[b] : 0;
    const int slot_n = b * (st.E + 1) + nom;
    const int slot_o = (a.mode == 0) ? slot_n : cand_slot(b, k, nom, st.E);
    const double *__restrict__ xbar = st.xs + (long)slot_n * st.x_stride;
    const double *__restrict__ lnom = (a.mode == 0) ? a.u0 : st.us + (long)slot_n * st.u_stride;
    double *__restrict__ xo = st.xs + (long)slot_o * st.x_stride;
    double *__restrict__ uo = st.us + (long)slot_o * st.u_stride;
    const int lsel = st.lsel[b];
    const double *__restrict__ Lb = st.L + (long)lsel * st.l_half + (long)b * N * LSTR;
    const double *__restrict__ dlb = st.dl + (long)lsel * st.dl_half + (long)b * N * USTR;

    double eps = 0.0;
    if (live && a.mode == 1) {
        eps = st.ls_eps[b];
        for (int q = 0; q < k; ++q) eps *= a.op.lambda;        // eps_k = eps * lambda^k by repeated multiplication (:530,:557)
    }
    // row j of [A|B] (time-invariant dynamics: f has no time argument in the reference)
    double zr[16];
    if (pb.model == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) zr[q] = (j < 12) ? pb.Zt[j * 16 + q] : 0.0;
    }
    const int jx = (j < 12) ? j : 11, ju = j & 3;          // clamped lane offsets: every load below is unconditional
    double x = 0.0;
    if (j < 12) x = (a.mode == 0) ? a.x0[j] : xbar[j];
    if (live && j < 12) xo[j] = x;
    double dmax = -INFINITY;
    bool dnan = false;
    int dom = 0;
    // software prefetch of the step-(t+1) operands (static load count: see load_tile)
    double n_xb = xbar[jx], n_l = lnom[ju], n_dl = dlb[ju];
    double n_L[12];
    if (a.mode == 1) {
#pragma unroll
        for (int q = 0; q < 12; ++q) n_L[q] = Lb[ju * 12 + q];
    }
    for (int t = 0; t < N; ++t) {
        const double c_xb = n_xb, c_l = n_l, c_dl = n_dl;
        double c_L[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) c_L[q] = n_L[q];
        {
            const int tn = (t + 1 < N) ? t + 1 : t;
            n_xb = xbar[(long)tn * XSTR + jx];
            n_l = lnom[(long)tn * USTR + ju];
            n_dl = dlb[(long)tn * USTR + ju];
            if (a.mode == 1) {
#pragma unroll
                for (int q = 0; q < 12; ++q) n_L[q] = Lb[(long)tn * LSTR + ju * 12 + q];
            }
        }
        double u = 0.0;
        if (a.mode == 1) {
            if (j < 12) shdx[row][j] = x - c_xb;
            WAVE_SYNC();
            {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                      // L_t (x_t - xbar_t)   (:82)
                    a0 = fma(c_L[q], shdx[row][q], a0);
                    a1 = fma(c_L[4 + q], shdx[row][4 + q], a1);
                    a2 = fma(c_L[8 + q], shdx[row][8 + q], a2);
                }
                const double lnew = c_l + eps * c_dl;                 // l + eps dl           (:509)
                u = lnew + ((a0 + a1) + a2);
                const double du = c_l - u;
                if (j < 4) shq[row][j] = du * du;
            }
        } else {
            u = c_l;
        }
        if (j < 12) shxu[row][j] = x;
        if (j < 4) shxu[row][12 + j] = u;
        WAVE_SYNC();
        double xn = 0.0;
        if (pb.model == 1) {
            double acc = 0.0, acc2 = 0.0, acc3 = 0.0, accb = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = fma(zr[q], shxu[row][q], acc);
                acc2 = fma(zr[4 + q], shxu[row][4 + q], acc2);
                acc3 = fma(zr[8 + q], shxu[row][8 + q], acc3);
                accb = fma(zr[12 + q], shxu[row][12 + q], accb);
            }
            acc = ((acc + acc2) + acc3) + accb;
            if (pb.kappa != 0.0) acc += pb.kappa * (x * x * x);
            xn = (j < 12) ? acc : 0.0;
        } else if (j < pb.n) {
            xn = powchk(x, pb.pl_a, dom) + powchk(shxu[row][12 + j], pb.pl_b, dom);
        }
        if (a.mode == 1 && j == 0) {
            const double dn = sqrt(shq[row][0] + shq[row][1] + shq[row][2] + shq[row][3]);
            if (dn != dn) dnan = true;                      // maximum() propagates NaN
            else if (dn > dmax) dmax = dn;
        }
        if (live) {
            if (j < 12) xo[(long)(t + 1) * XSTR + j] = xn;
            if (j < 4) uo[(long)t * USTR + j] = u;
        }
        x = xn;
        WAVE_SYNC();
    }
    // any lane of the row saw a DomainError?
    const unsigned long long bal = __ballot(dom != 0);
    const int rowdom = ((bal >> (row * 16)) & 0xFFFFull) != 0;
    if (live && j == 0) {
        if (a.mode == 1) {
            st.d_c[c] = dnan ? NAN : dmax;
            st.flag_c[c] = rowdom ? 2 : 0;
        } else if (rowdom) {
            st.status[b] = 4;               // RAT_ST_DOMAIN
            st.value[b] = INFINITY;
        }
    }
}

void launch_rollout(const RolloutArgs &a, hipStream_t s) {
    const int ncand = (a.mode == 0) ? a.st.B : a.st.B * a.st.E;
    if (ncand <= 0) return;
    hipLaunchKernelGGL(rollout_kernel, dim3((ncand + 3) / 4), dim3(64), 0, s, a);
}

#endif  // PART_ROLL
// =====================================================================================================
// rollin: fused simulate_dynamics + approximate_model for the solver's hot loop (ileqg.jl:62-87 then :258-322).
// ONE wavefront per trajectory.  The recursion x_{t+1} = [A|B][x_t; u_t], u_t = l_t + eps dl_t + L_t (x_t - xbar_t) is a
// chain of tiny matrix-vector products whose operands would have to be broadcast through LDS every step; instead the
// vectors live in "B-form" -- register s of lane (g, j) holds component 4 s + g, the B-operand (and, read the other way,
// the A-operand) slice s of v_mfma_f64_16x16x4_f64 with the vector replicated across the 16 columns -- and every product
// is issued to the matrix pipe with a constant (or prefetched) operand on the other side:
//   L_t dx_t           A = L_t column slices,  B = dx in B-form          -> row g of D = (L dx)_g   = B-form slice 3 of [x; u]
//   [A|B][x; u]        A = [A|B] column slices, B = [x; u] in B-form     -> register r of D = x_{t+1} in B-form: the recursion
//                                                                           closes without a single cross-lane move
//   C [x; u]           A = C column slices (C symmetric: the register image of C), B = [x; u] in B-form
//                                                                        -> register r of D = (C [x;u])_{4r+g} = [c_x | c_u] - lin in B-form
// The critical path per step is 3 + 1 dependent MFMAs (the first three [A|B] slices do not wait for u).  Off the critical
// path a B-form vector is "packed" -- lane (g, s), s < 4, selects its register s, so 16 lanes hold the 16 components -- and
// [x_t; u_t], [c_x | c_u] and c leave through per-lane store addresses.  No LDS and no fence inside the time loop: the
// whole group of RD steps is one basic block of pure data flow, which lets the scheduler run one step's cost gradient and
// tile stores under the recursion of the next (an in-order wavefront has no other source of overlap).  x_t, u_t never make a round trip through HBM between the two
// reference functions.  rollout_kernel + linearize_kernel (operator entry points) compute the same quantities.
// =====================================================================================================
// CTV: time-varying cost tables (LQ family).  A template parameter, not a branch: a conditional per-step table load would put
// a path-dependent number of loads between the prefetch and the tile stores and collapse every counted vmcnt wait.
// STAGE (closed loop, N <= ROLLIN_NST): the operands of the whole trajectory -- L, xbar, l, dl: 27 KB -- are copied into LDS before the
// time loop (57 loads in flight at once), so the loop issues no global loads at all and its tile stores never meet a vmcnt wait.
// SEP (operand loads in the loop only): keep the steps of a group apart in the instruction schedule (see the time loop).
// shxu: 16 doubles of this wavefront's LDS (terminal tile); stg: STG_DOUBLES of LDS shared by the waves of the workgroup (STAGE), else null
// NOTILE (line-search candidates of the speculative path, LQ family): the step record keeps only [c_x | c_u], c and its zero pair; the
// sweeps that evaluate the candidate form f_x | f_u and the cost Hessian of step t from (x_t, u_t) and the problem tables themselves
// (load_tile<.., FLY>); the gain sweeps that may read an accepted candidate's records are fly sweeps too, so the rest of a record is never
// written -- the reference keeps approximate_model's result only for the trajectory it accepts (ileqg.jl:514-555), this path for none.
template <int MODEL, int MODE, bool CTV, bool STAGE = false, bool SEP = true, int PF = ROLLIN_PREFETCH, bool NOTILE = false>
__device__ __forceinline__ void rollin_body(const RolloutArgs &a, const int c, double *const shxu, double *const stg = nullptr) {
    int lane_ = threadIdx.x & 63;        // (rollin_stage_kernel runs the E candidates of a sample as the waves of one workgroup)
    asm volatile("" : "+v"(lane_));      // opaque per phase (see sweep_body)
    const int l = lane_, j = l & 15, g = l >> 4;
#ifdef RAT_DIAG
    const unsigned long long dg_entry = __builtin_readcyclecounter();
    unsigned long long dg_loop0 = 0;
#endif
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    BODY_MARK(a.dump, 12 + 4 * MODE + 0);
    int b, k = 0;
    if (MODE == 0) b = c;
    else { b = c / st.E; k = c - b * st.E; }
    // (per-sample scalars fetched before the first test on any of them: see sweep_body)
    const int v_stat = st.status[b], v_act = st.ls_active[b], v_nom = st.slot_nom[b], v_lsel = st.lsel[b];
    const int s_stat = wave_uniform(v_stat), s_act = wave_uniform(v_act), nom = wave_uniform(v_nom), lsel = wave_uniform(v_lsel);
    const double eps_in = st.ls_eps[b];
    if (MODE == 0) { if (s_stat != ST_RUNNING) return; }
    else { if (!s_act) return; }

    const int slot_n = b * (st.E + 1) + nom;
    const int slot_o = (MODE == 0) ? slot_n : cand_slot(b, k, nom, st.E);
    const double *__restrict__ xbar = st.xs + (long)slot_n * st.x_stride;
    const double *__restrict__ lnom = (MODE == 0) ? a.u0 : st.us + (long)slot_n * st.u_stride;
    double *__restrict__ xo = st.xs + (long)slot_o * st.x_stride;
    double *__restrict__ uo = st.us + (long)slot_o * st.u_stride;
    double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot_o) * st.tile_stride;
    const double *__restrict__ Lb = st.L + (long)lsel * st.l_half + (long)b * N * LSTR;
    const double *__restrict__ dlb = st.dl + (long)lsel * st.dl_half + (long)b * N * USTR;
    constexpr bool lq = (MODEL == 1);
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};

    double eps = 0.0;
    if (MODE == 1) {
        eps = eps_in;
        for (int q = 0; q < k; ++q) eps *= a.op.lambda;        // eps_k = eps * lambda^k by repeated multiplication (:530,:557)
    }
    // per-lane constants (dynamics are time-invariant; cost tables only when !CTV).  The tile is a register image
    // (layout.h): lane l owns element 64 r + l of Z and of C = [[Q, 0], [P, R]].
    double zA[4] = {0, 0, 0, 0};        // A-operand slice s of [A|B]: lane (g, j) holds [A|B][j][4 s + g], rows j >= 12 zero
    double cf[4] = {0, 0, 0, 0};        // register r of the C image: lane (g, j) holds C[4 r + g][j] = C[j][4 r + g] (A-operand slice r)
    double es[4];                       // B-operand slice s of the by-j transposition (x_N only): 1 where j == 4 s + g
    double zt0 = 0, zt1 = 0, zt2 = 0, cq0 = 0, cq1 = 0, cq2 = 0, cpr = 0, clin = 0, cq00 = 0;
    const double mq = (j < 12) ? 1.0 : 0.0;                     // rows 0..11 of C: columns 12..15 are dead slots, written as 0
    const int jx = (j < 12) ? j : 11, j3 = j & 3;
    const int pkc = 4 * j3 + g;                                 // component this lane carries in packed form (lanes j < 4)
#pragma unroll
    for (int s = 0; s < 4; ++s) es[s] = (j == 4 * s + g) ? 1.0 : 0.0;
    const double dgz[3] = {es[0], es[1], es[2]};                // 1 on the lane that holds the diagonal element of row 4 r + g of f_x
    if (lq) {
#pragma unroll
        for (int s = 0; s < 4; ++s) zA[s] = pb.Zt[jx * 16 + 4 * s + g] * mq;
        zt0 = pb.Zt[l]; zt1 = pb.Zt[64 + l]; zt2 = pb.Zt[128 + l];
        if (!CTV) {
#pragma unroll
            for (int s = 0; s < 4; ++s) cf[s] = pb.Ctab[64 * s + l];
            cq0 = cf[0] * mq; cq1 = cf[1] * mq; cq2 = cf[2] * mq; cpr = cf[3];
            clin = pb.lin[pkc];
            cq00 = pb.q0[0];
        }
    }
    // one store per step writes x_t (lanes 0..11), u_t (lanes 12..15) and, from the idle lanes, 0.0 to the record's pad slot:
    // per-lane base and stride instead of a per-step address select (which compiles to a divergent branch)
    // [x_t; u_t] packed: lane (g, s), s < 4, holds component 4 s + g (B-form register s of that lane)
    double *const pxu = (j < 3) ? xo + 4 * j + g : (j == 3 ? uo + g : tile0 + TS_PAD);
    const long sxu = (j < 3) ? XSTR : (j == 3 ? USTR : TSTRIDE);
    // [c_x | c_u] packed the same way; lane 4 carries c, the other idle lanes 0.0 for the pad slot
    const int qoff = (j < 4) ? TS_QR + 4 * j + g : (l == 4 ? TS_q : TS_PAD);
    const int c34 = TS_REG(3, l), r5 = TS_REG(5, l);            // C rows 0..11: live lanes compact, dead lanes (j >= 12) to the zero pair
    const double pm[4] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0, j == 3 ? 1.0 : 0.0};
    const double m_j4 = (j < 4) ? 1.0 : 0.0, m_l4 = (l == 4) ? 1.0 : 0.0;
    // QB (LQ family, time-invariant cost): the cost gradients of a GROUP of RD steps in one set of four MFMAs.  C [x_t; u_t] is off the
    // recursion's dependency chain, and in B-form a vector is replicated over the 16 columns of the B operand -- so column d of a second
    // operand set (xq) collects [x_t; u_t] of step t0 + d as the group goes, and at its end C xq gives [c_x | c_u] - lin of all RD steps:
    // 4 MFMAs per group instead of 4 per step (RD = 5: 7.8 instead of 11 per closed-loop step).  Each column of an MFMA is formed from
    // its own column of B in a fixed order, and the per-lane expressions are those of the per-step form (row sums in its order: see
    // rollin_multi_kernel), so the records hold the same bits (the block kernel's rolllin_body keeps the per-step form: tested against it).
    constexpr bool QB = lq && !CTV;
    double xq[4] = {0.0, 0.0, 0.0, 0.0}, lin4[4] = {0.0, 0.0, 0.0, 0.0};
    if (QB) {
#pragma unroll
        for (int s = 0; s < 4; ++s) lin4[s] = pb.lin[4 * s + g];
    }
    double xb[3];                                               // x_t in B-form
#pragma unroll
    for (int s = 0; s < 3; ++s) xb[s] = (MODE == 0) ? a.x0[4 * s + g] : xbar[4 * s + g];
    double dmax = -INFINITY;
    bool dnan = false;
    int dom = 0;
    // Software prefetch of the step operands, RD-1 steps ahead, in RD rotating register sets (the time loop is unrolled by RD).
    // gfx9 counts loads and stores in ONE in-order vmcnt, so waiting for a prefetched operand also waits for every store
    // issued before that prefetch.  With the prefetch of step t+RD-1 issued at the top of step t, the wait at the top of
    // step t+RD-1 leaves the tile stores of the last RD-1 steps in flight (vmcnt = stores + (RD-2)(loads + stores) + loads),
    // which is what keeps the HBM write pipe full: a distance of one drains the tile stream behind a single step of
    // arithmetic and the wave then idles a write latency (~1 us) per step.  The count has to hold on every path into the
    // loop: (1) all stores of a step are unconditional (idle lanes write 0.0 to the record's pad slot), (2) each prefetch of
    // the prologue is followed by as many (pad) stores as a step issues, (3) the loads are unconditional (clamped step index).
    // A register set is refilled at the top of the step AFTER the one that consumed it: its old value is dead by then, so
    // the loop-carried sets need no copies (a copy of a just-loaded register would wait for the load and drain the queue).
    constexpr int RD = (MODEL == 1) ? PF : 2;     // (power-law family: pow() expansions are large -- keep its loop short)
    constexpr int kStoresPerStep = 6;
    struct StepIn { double l, dl, xb[3], La[3]; };
    StepIn buf[RD];
    constexpr bool staged = STAGE;                // (open loop: the nominal controls alone -- its time loop then issues no global load either)
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    // (open loop: only the controls, at the start of the area -- STG_CU * 64 doubles are enough for a kernel that runs nothing else)
    double *const sL = stg, *const sX = stg + (staged ? cL * 64 : 0), *const sl = (MODE == 0) ? stg : sX + (staged ? cX * 64 : 0),
                 *const sdl = sl + (staged ? cU * 64 : 0);
    if (staged && MODE == 0) {
        double tl[cU];
#pragma unroll
        for (int q = 0; q < cU; ++q) { const int e = 64 * q + l; tl[q] = lnom[(e < N * USTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cU; ++q) sl[64 * q + l] = tl[q];
        WAVE_SYNC();
    }
    if (staged && MODE == 1) {
        double tL[cL], tX[cX], tl[cU], tdl[cU];
#pragma unroll
        for (int q = 0; q < cL; ++q) { const int e = 64 * q + l; tL[q] = Lb[(e < N * LSTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cX; ++q) { const int e = 64 * q + l; tX[q] = xbar[(e < (N + 1) * XSTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cU; ++q) { const int e = 64 * q + l; tl[q] = lnom[(e < N * USTR) ? e : 0]; tdl[q] = dlb[(e < N * USTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cL; ++q) sL[64 * q + l] = tL[q];
#pragma unroll
        for (int q = 0; q < cX; ++q) sX[64 * q + l] = tX[q];
#pragma unroll
        for (int q = 0; q < cU; ++q) { sl[64 * q + l] = tl[q]; sdl[64 * q + l] = tdl[q]; }
        WAVE_SYNC();
    }
    auto issue = [&](StepIn &in, const int tq) {
        const int tn = (tq < N) ? tq : N - 1;
        in.dl = 0.0;
#pragma unroll
        for (int s = 0; s < 3; ++s) in.xb[s] = in.La[s] = 0.0;
        if (staged) {
            in.l = sl[tn * USTR + g];
            if (MODE == 1) {
                in.dl = sdl[tn * USTR + g];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    in.xb[s] = sX[tn * XSTR + 4 * s + g];
                    in.La[s] = sL[tn * LSTR + j3 * 12 + 4 * s + g];
                }
            }
            return;
        }
        in.l = lnom[(long)tn * USTR + g];
        if (MODE == 1) {
            in.dl = dlb[(long)tn * USTR + g];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                in.xb[s] = xbar[(long)tn * XSTR + 4 * s + g];
                in.La[s] = Lb[(long)tn * LSTR + j3 * 12 + 4 * s + g];   // A-operand slice s of L_t: rows >= 4 of the product are never read
            }
        }
    };
#pragma unroll
    for (int d = 0; d < RD - 1; ++d) {
        issue(buf[d], d);
        if (staged) continue;                                   // operands come from LDS: no vmcnt wait in the loop to calibrate
#pragma unroll
        for (int q = 0; q < kStoresPerStep; ++q) {              // (distinct pad slots: identical stores would be merged away)
            const int tq = d * kStoresPerStep + q;
            tile0[(long)((tq < N) ? tq : N - 1) * TSTRIDE + TS_PAD] = 0.0;
        }
    }
    buf[RD - 1].l = buf[RD - 1].dl = 0.0;       // (defined on entry; refilled at the top of step 0)
#pragma unroll
    for (int s = 0; s < 3; ++s) buf[RD - 1].xb[s] = buf[RD - 1].La[s] = 0.0;
    DIAG_DECL
#ifdef RAT_DIAG
    dg_loop0 = __builtin_readcyclecounter();
#endif
    auto step = [&](const int t, const StepIn &cur, const int d, const bool last) {      // d: position in the group; last: last step before a flush
        DIAG_START();
        const double c_l = cur.l, c_dl = cur.dl;
        const double c_xb[3] = {cur.xb[0], cur.xb[1], cur.xb[2]}, c_La[3] = {cur.La[0], cur.La[1], cur.La[2]};
        double *__restrict__ tp = tile0 + (long)t * TSTRIDE;
        if (CTV) {
            const double *__restrict__ C = pb.Ctab + (long)t * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) cf[s] = C[64 * s + l];
            cq0 = cf[0] * mq; cq1 = cf[1] * mq; cq2 = cf[2] * mq; cpr = cf[3];
            clin = pb.lin[(long)t * 16 + pkc];
            cq00 = pb.q0[t];
        }
        // ---- x_{t+1} = f(x_t, u_t): the x-part of [A|B][x; u] does not wait for the feedback control -----------------
        d4 xa = zero4;
        if (lq) {
            xa = MFMA(zA[0], xb[0], xa);
            xa = MFMA(zA[1], xb[1], xa);
            xa = MFMA(zA[2], xb[2], xa);
        }
        double u = c_l;                                             // u_g on the lanes of row g (B-form slice 3)
        if (MODE == 1) {
            d4 fb = MFMA(c_La[0], xb[0] - c_xb[0], zero4);          // L_t (x_t - xbar_t)   (:82)
            fb = MFMA(c_La[1], xb[1] - c_xb[1], fb);
            fb = MFMA(c_La[2], xb[2] - c_xb[2], fb);
            const double lnew = c_l + eps * c_dl;                   // l + eps dl           (:509)
            u = lnew + fb[0];
            // d = maximum(norm(l_t - u_t))  (:517-519): sqrt is monotone, so the maximum is taken over the squared norms and
            // rooted once after the loop (same bits); maximum() propagates NaN
            const double du = c_l - u, dsq = du * du;
            const double dn2 = ((readlane_f64(dsq, 0) + readlane_f64(dsq, 16)) + readlane_f64(dsq, 32)) + readlane_f64(dsq, 48);
            dnan |= (dn2 != dn2);
            dmax = (dn2 > dmax) ? dn2 : dmax;
        }
        DIAG_STAMP(0, u);
        double xn[3] = {0.0, 0.0, 0.0};
        if (lq) {
            xa = MFMA(zA[3], u, xa);
            // + kappa x^3, branch-free (kappa = 0 adds 0): the whole step stays ONE basic block, so the scheduler can run the
            // off-path chains (transposition, C [x;u], tile stores) of one step under the recursion of the next
#pragma unroll
            for (int r = 0; r < 3; ++r) xn[r] = xa[r] + pb.kappa * (xb[r] * xb[r] * xb[r]);
        } else if (g < pb.n) {
            xn[0] = powchk(xb[0], pb.pl_a, dom) + powchk(u, pb.pl_b, dom);
        }
        DIAG_STAMP(1, xn[0]);
        // ---- [x_t; u_t] packed on 16 lanes -> one store (idle lanes: 0.0 to the pad slot) ---------------------------------
        // (0/1 multipliers instead of selects: a select between computed values compiles to a divergent branch, and one branch
        //  splits the basic block the scheduler needs whole; exactly one term is non-zero, so the sum is exact)
        const double pk = ((xb[0] * pm[0] + xb[1] * pm[1]) + xb[2] * pm[2]) + u * pm[3];
        pxu[(long)t * sxu] = pk;
        // ---- tile of step t: approximate_model at (x_t, u_t)   (ileqg.jl:294-313) ---------------------
        if (lq) {
            // f_x = A + diag(3 kappa x^2) | f_u = B: the diagonal element of row 4 r + g sits on lane (g, 4 r + g), which holds x_{4r+g}
            if (!NOTILE) {
                const double z0 = fx_diag(zt0, dgz[0], pb.kappa, xb[0]);
                const double z1 = fx_diag(zt1, dgz[1], pb.kappa, xb[1]);
                const double z2 = fx_diag(zt2, dgz[2], pb.kappa, xb[2]);
                double2 *__restrict__ t2 = reinterpret_cast<double2 *>(tp);
                t2[l] = make_double2(z0, z1);
                t2[64 + l] = make_double2(z2, cpr);
                *reinterpret_cast<double2 *>(tp + c34) = make_double2(cq0, cq1);     // dead lanes: (0, 0) to the zero pair
                tp[r5] = cq2;
            }
            if (QB) {
                // column d of xq <- [x_t; u_t]; the last step before a flush also fills the columns behind it (they then mirror a live
                // column: every store of the flush stays unconditional, duplicates carry identical values to identical addresses)
                const bool me = (j == d) || (last && j > d);
#pragma unroll
                for (int s = 0; s < 3; ++s) xq[s] = me ? xb[s] : xq[s];
                xq[3] = me ? u : xq[3];
            } else {
                d4 cx = MFMA(cf[0], xb[0], zero4);                      // C [x;u] in B-form
                cx = MFMA(cf[1], xb[1], cx);
                cx = MFMA(cf[2], xb[2], cx);
                cx = MFMA(cf[3], u, cx);
                const double acc = ((cx[0] * pm[0] + cx[1] * pm[1]) + cx[2] * pm[2]) + cx[3] * pm[3];    // packed (lanes j < 4), 0 elsewhere
                // c = [x;u]' (1/2 C [x;u] + lin) + q0  (:296): 16 packed terms, summed per row and then over the four rows
                const double w = row_sum16(cost_term(pk, acc, clin));    // pk = 0 on the idle lanes
                const double part = ((readlane_f64(w, 0) + readlane_f64(w, 16)) + readlane_f64(w, 32)) + readlane_f64(w, 48);
                tp[qoff] = fma(m_l4, part + cq00, m_j4 * (acc + clin));  // [c_x | c_u] = C [x;u] + [qv;rv]  (:297,:299), c, pad (0.0)
            }
        } else {
            // power-law family (n == m <= 4): every derivative is diagonal, and row g's entries sit on the lanes of row g
            double val = 0.0, cv = 0.0;
            if (g < pb.n) {
                if (j == g) {
                    val = pb.pl_a * powchk(xb[0], pb.pl_a - 1.0, dom);
                    cv = pb.pl_cx * pb.pl_p * (pb.pl_p - 1.0) * powchk(xb[0], pb.pl_p - 2.0, dom);
                } else if (j == 12 + g) val = pb.pl_b * powchk(u, pb.pl_b - 1.0, dom);
            }
            double2 *__restrict__ t2 = reinterpret_cast<double2 *>(tp);
            double rv = 0.0;
            if (j == 12 + g) rv = (g < pb.m) ? pb.pl_cu * pb.pl_pu * (pb.pl_pu - 1.0) * powchk(u, pb.pl_pu - 2.0, dom) : 1.0;
            t2[l] = make_double2(val, 0.0);
            t2[64 + l] = make_double2(0.0, rv);
            *reinterpret_cast<double2 *>(tp + c34) = make_double2(cv, 0.0);      // cv != 0 only on the diagonal lane j == g (live)
            tp[r5] = 0.0;
            double part = 0.0, qv = 0.0;                            // packed: x_g on lane (g, 0), u_g on lane (g, 3)
            if (j == 0 && g < pb.n) {
                qv = pb.pl_cx * pb.pl_p * powchk(pk, pb.pl_p - 1.0, dom);
                part = pb.pl_cx * powchk(pk, pb.pl_p, dom);
            } else if (j == 3 && g < pb.m) {
                qv = pb.pl_cu * pb.pl_pu * powchk(pk, pb.pl_pu - 1.0, dom);
                part = pb.pl_cu * powchk(pk, pb.pl_pu, dom);
            }
            part = wave_sum(part);
            tp[qoff] = (j < 4) ? qv : (l == 4 ? part : 0.0);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) xb[r] = xn[r];
        DIAG_STAMP(2, xb[0]);
    };
    // the cost-gradient rows and costs of steps t0 .. t0 + cnt - 1 (columns 0 .. cnt - 1 of xq; the columns behind mirror the last one)
    auto flush = [&](const int t0, const int cnt) {
        d4 cx = MFMA(cf[0], xq[0], zero4);                          // C [x;u] of every step of the group
        cx = MFMA(cf[1], xq[1], cx);
        cx = MFMA(cf[2], xq[2], cx);
        cx = MFMA(cf[3], xq[3], cx);
        double *__restrict__ rp = tile0 + (long)(t0 + ((j < cnt) ? j : cnt - 1)) * TSTRIDE;
#pragma unroll
        for (int s = 0; s < 4; ++s) rp[TS_QR + 4 * s + g] = cx[s] + lin4[s];                    // [c_x | c_u] = C [x;u] + [qv;rv]  (:297,:299)
        // c = [x;u]' (1/2 C [x;u] + lin) + q0  (:296), summed in the per-step form's order: (a0 + a2) + (a1 + a3) per row, rows in order
        const double a0 = cost_term(xq[0], cx[0], lin4[0]), a1 = cost_term(xq[1], cx[1], lin4[1]);
        const double a2 = cost_term(xq[2], cx[2], lin4[2]), a3 = cost_term(xq[3], cx[3], lin4[3]);
        double wr[4];
        rows_bcast((a0 + a2) + (a1 + a3), wr);
        const double part = ((wr[0] + wr[1]) + wr[2]) + wr[3];
        rp[(g == 0) ? TS_q : TS_PAD + (g & 1)] = (g == 0) ? part + cq00 : 0.0;                  // row 0: c; rows 1..3: the record's zero pair
    };
    // single-exit main loop over whole groups of RD steps (a second exit would put a path from the middle of the group
    // back to the loop header into the control-flow graph and cap the header's vmcnt at that path's count), then the tail
    int t0 = 0;
    BODY_MARK(a.dump, 12 + 4 * MODE + 1);
    for (; t0 + RD <= N; t0 += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            // the group of RD steps is one basic block.  With the operands staged in LDS the scheduler may interleave the steps
            // freely.  With operand loads in the loop it sinks every store of the group below the loads of the last step, which
            // collapses the counted-vmcnt pipeline above: at one wave per SIMD (E = 1) that costs 15 % and nothing may cross a step
            // boundary (SEP); with two waves per SIMD (E > 1) the other wave fills the gaps and the freer schedule measured 5 % faster.
            if (!staged && SEP) __builtin_amdgcn_sched_barrier(0);
            issue(buf[(d + RD - 1) % RD], t0 + d + RD - 1);
            step(t0 + d, buf[d], d, d == RD - 1);
        }
        if (QB) flush(t0, RD);
    }
    {
        const int nt = N - t0;                                  // N mod RD steps: their operands are already in buf[0..]
#pragma unroll
        for (int d = 0; d < RD - 1; ++d)
            if (d < nt) step(t0 + d, buf[d], d, d == nt - 1);
        if (QB && nt > 0) flush(t0, nt);
    }
#ifdef RAT_DIAG
    if (l == 0 && blockIdx.x < 8 && a.dump) {
        for (int q = 0; q < 3; ++q) a.dump[64 + blockIdx.x * 8 + q] = (double)dg_acc[q];
        a.dump[64 + blockIdx.x * 8 + 3] = (double)(dg_loop0 - dg_entry);                       // prologue
        a.dump[64 + blockIdx.x * 8 + 5] = (double)dg_gap;                                      // between end-of-step stamp and next start
        a.dump[64 + blockIdx.x * 8 + 4] = (double)(__builtin_readcyclecounter() - dg_entry);  // entry .. end of time loop
    }
#endif
    BODY_MARK(a.dump, 12 + 4 * MODE + 2);
    // ---- x_N and the terminal tile: h, h_x, h_xx at x_N   (ileqg.jl:314-316) -------------------------------
    {
        d4 tj = MFMA(xb[0], es[0], zero4);
        tj = MFMA(xb[1], es[1], tj);
        tj = MFMA(xb[2], es[2], tj);
        const double x = tj[0];                                     // x_N by lane j (lanes j >= 12: 0)
        if (l < 12) { xo[(long)N * XSTR + l] = x; shxu[l] = x; }
        WAVE_SYNC();
        double *__restrict__ tp = tile0 + (long)N * TSTRIDE;
        if (lq) {
            for (int e = l; e < 144; e += 64) tp[TT_Q + e] = pb.Qf[e];
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = fma(pb.Qf[jx * 12 + q], shxu[q], acc);
            const double qvf = pb.qvf[jx];
            if (l < 12) tp[TT_QV + l] = acc + qvf;
            const double part = row_sum16((j < 12) ? cost_term(shxu[jx], acc, qvf) : 0.0);
            if (l == 0) tp[TT_q] = part + pb.q0f;
        } else {
            for (int e = l; e < 144; e += 64) tp[TT_Q + e] = 0.0;
            if (l < 12) tp[TT_QV + l] = 0.0;
            if (l == 0) tp[TT_q] = pb.pl_h;
        }
        WAVE_SYNC();                                                // shxu may be rewritten by the next phase of a fused solve
    }
    const bool anydom = __ballot(dom != 0) != 0ull;
    if (l == 0) {
        if (MODE == 1) {
            st.d_c[c] = dnan ? NAN : sqrt(dmax);
            st.flag_c[c] = anydom ? 2 : 0;
        } else if (anydom) {
            st.status[b] = 4;               // RAT_ST_DOMAIN
            st.value[b] = INFINITY;
        }
    }
    BODY_MARK(a.dump, 12 + 4 * MODE + 3);
}

// =====================================================================================================
// rollin split over TWO wavefronts of a workgroup (solve_block_kernel, LQ family, N <= ROLLIN_NST): the recursion and the
// linearisation of rollin_body on different SIMDs.
//   rollrec_body (candidate wave): x_{t+1} = [A|B][x_t; u_t], u_t = l_t + eps dl_t + L_t (x_t - xbar_t) -- 7 MFMAs per step (4 open
//                loop) and nothing else: no tile, no store to HBM.  Operands come from LDS (staged once), [x_t; u_t] goes to an LDS
//                trajectory buffer (16 doubles per step) followed by a progress word.
//   rolllin_body (the gain wave, idle during rollouts otherwise): waits for step t, reads [x_t; u_t] from LDS and does everything
//                else rollin_body does per step -- f_x | f_u, C [x;u] (4 MFMAs), c, the six record stores, the x / u history.
// Same expressions as rollin_body (the paths are tested bit for bit).  A rollout then takes ~50 x 560 cycles (both halves are bound
// by their MFMAs: 7 x 64 and 4 x 64 + stores) instead of 62 k (open loop) / 85 k (closed loop): -127 k of the 760 k-cycle critical
// path of a 2-iteration solve at one wave per SIMD.
// Hand-off: LDS operations of one CU execute in arrival order, and a wave's own LDS instructions are issued in program order, so
// "data written, then progress word written" by the producer and "progress word read, then data read" by the consumer need no fence
// beyond the compiler's (wavefront scope).  The progress word counts up over the whole solve (epoch + t + 1): no reset, no ABA.
// =====================================================================================================
#ifdef RAT_DIAG_PHASES
#define ACL_MARK(dump_, wave_, slot_, val_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8 && (dump_)) \
        (dump_)[2048 + blockIdx.x * 64 + (wave_) * 16 + (slot_)] = (double)(val_); } while (0)
#else
#define ACL_MARK(dump_, wave_, slot_, val_) do {} while (0)
#endif
#define XU_REC 20                                   /* doubles per step: [x_t; u_t] packed (16) + (l_t - u_t)^2 per control (4) */
#define XU_DOUBLES ((ROLLIN_NST + 1) * XU_REC + 64) /* t = 0..N, + dump slots of the idle lanes */

// true when this sample takes part in the rollout phase MODE (both waves evaluate the same words: they agree)
template <int MODE>
__device__ __forceinline__ bool rollout_active(const StateDev &st, const int b, int &nom, int &lsel, double &eps_in) {
    const int v_stat = xld(&st.status[b]), v_act = xld(&st.ls_active[b]), v_nom = xld(&st.slot_nom[b]), v_lsel = xld(&st.lsel[b]);
    const int s_stat = wave_uniform(v_stat), s_act = wave_uniform(v_act);
    nom = wave_uniform(v_nom); lsel = wave_uniform(v_lsel);
    eps_in = xld(&st.ls_eps[b]);
    return (MODE == 0) ? (s_stat == ST_RUNNING) : (s_act != 0);
}

// HELP (the workgroup has spare linearise waves: one sample per CU): the squared control steps go to the trajectory buffer and the
// linearise waves take their maximum (d_acc), so that the recursion wave's step is nothing but its MFMA chain.
template <int MODE, bool HELP>
__device__ __forceinline__ void rollrec_body(const RolloutArgs &a, const int b, double *const stg, double *const xu, int *const prog, const int epoch,
                                             unsigned long long *const d_acc) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l = lane_, j = l & 15, g = l >> 4;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    int nom, lsel;
    double eps_in;
    if (!rollout_active<MODE>(st, b, nom, lsel, eps_in)) return;
    const int slot_n = b * (st.E + 1) + nom;
    const double *__restrict__ xbar = st.xs + (long)slot_n * st.x_stride;
    const double *__restrict__ lnom = (MODE == 0) ? a.u0 : st.us + (long)slot_n * st.u_stride;
    const double *__restrict__ Lb = st.L + (long)lsel * st.l_half + (long)b * N * LSTR;
    const double *__restrict__ dlb = st.dl + (long)lsel * st.dl_half + (long)b * N * USTR;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const double eps = (MODE == 1) ? eps_in : 0.0;                  // candidate 0: eps_k = eps lambda^0
    const double mq = (j < 12) ? 1.0 : 0.0;
    const int jx = (j < 12) ? j : 11, j3 = j & 3;
    double zA[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) zA[s] = pb.Zt[jx * 16 + 4 * s + g] * mq;
    const double pm[4] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0, j == 3 ? 1.0 : 0.0};
    // packed [x; u]: lane (g, s), s < 4, holds component 4 s + g; lane (g, 4): (l_g - u_g)^2 (HELP); idle lanes: dump
    const int xoff = (j < 4) ? 4 * j + g : ((HELP && j == 4) ? 16 + g : (ROLLIN_NST + 1) * XU_REC + l);
    const int xstep = (j < 4 || (HELP && j == 4)) ? XU_REC : 0;
    const double m_4 = (j == 4) ? 1.0 : 0.0;
    if (HELP && MODE == 1 && l == 0) { d_acc[0] = 0ull; d_acc[1] = 0ull; }     // max of the squared step norms (bits of a double >= +0), NaN flag
    double xb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) xb[s] = (MODE == 0) ? a.x0[4 * s + g] : xbar[4 * s + g];
    // operands of the whole trajectory into LDS (rollin_body's STAGE layout; open loop: the controls only)
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    double *const sL = stg, *const sX = stg + cL * 64, *const sl = sX + cX * 64, *const sdl = sl + cU * 64;
    {
        double tl[cU];
#pragma unroll
        for (int q = 0; q < cU; ++q) { const int e = 64 * q + l; tl[q] = lnom[(e < N * USTR) ? e : 0]; }
        if (MODE == 1) {
            double tL[cL], tX[cX], tdl[cU];
#pragma unroll
            for (int q = 0; q < cL; ++q) { const int e = 64 * q + l; tL[q] = xld(&Lb[(e < N * LSTR) ? e : 0]); }
#pragma unroll
            for (int q = 0; q < cX; ++q) { const int e = 64 * q + l; tX[q] = xbar[(e < (N + 1) * XSTR) ? e : 0]; }
#pragma unroll
            for (int q = 0; q < cU; ++q) { const int e = 64 * q + l; tdl[q] = xld(&dlb[(e < N * USTR) ? e : 0]); }
#pragma unroll
            for (int q = 0; q < cL; ++q) sL[64 * q + l] = tL[q];
#pragma unroll
            for (int q = 0; q < cX; ++q) sX[64 * q + l] = tX[q];
#pragma unroll
            for (int q = 0; q < cU; ++q) sdl[64 * q + l] = tdl[q];
        }
#pragma unroll
        for (int q = 0; q < cU; ++q) sl[64 * q + l] = tl[q];
        WAVE_SYNC();
    }
    double dmax = -INFINITY;
    bool dnan = false;
    // operands of step t + 1 are read from LDS during step t (the fence before the progress word would otherwise pin them to their step)
    double n_l = sl[g], n_dl = 0.0, n_xb[3] = {0, 0, 0}, n_La[3] = {0, 0, 0};
    if (MODE == 1) {
        n_dl = sdl[g];
#pragma unroll
        for (int s = 0; s < 3; ++s) { n_xb[s] = sX[4 * s + g]; n_La[s] = sL[j3 * 12 + 4 * s + g]; }
    }
    for (int t = 0; t < N; ++t) {
        const double c_l = n_l;
        const int tn = (t + 1 < N) ? t + 1 : t;
        n_l = sl[tn * USTR + g];
        d4 xa = zero4;
        xa = MFMA(zA[0], xb[0], xa);
        xa = MFMA(zA[1], xb[1], xa);
        xa = MFMA(zA[2], xb[2], xa);
        double u = c_l;
        double dsq = 0.0;
        if (MODE == 1) {
            const double c_dl = n_dl;
            const double c_xb[3] = {n_xb[0], n_xb[1], n_xb[2]}, c_La[3] = {n_La[0], n_La[1], n_La[2]};
            n_dl = sdl[tn * USTR + g];
#pragma unroll
            for (int s = 0; s < 3; ++s) { n_xb[s] = sX[tn * XSTR + 4 * s + g]; n_La[s] = sL[tn * LSTR + j3 * 12 + 4 * s + g]; }
            d4 fb = MFMA(c_La[0], xb[0] - c_xb[0], zero4);          // L_t (x_t - xbar_t)   (:82)
            fb = MFMA(c_La[1], xb[1] - c_xb[1], fb);
            fb = MFMA(c_La[2], xb[2] - c_xb[2], fb);
            const double lnew = c_l + eps * c_dl;                   // l + eps dl           (:509)
            u = lnew + fb[0];
            const double du = c_l - u;                              // d = maximum(norm(l_t - u_t))  (:517-519), as rollin_body
            dsq = du * du;
            if (!HELP) {
                const double dn2 = ((readlane_f64(dsq, 0) + readlane_f64(dsq, 16)) + readlane_f64(dsq, 32)) + readlane_f64(dsq, 48);
                dnan |= (dn2 != dn2);
                dmax = (dn2 > dmax) ? dn2 : dmax;
            }
        }
        xa = MFMA(zA[3], u, xa);
        const double pk = ((xb[0] * pm[0] + xb[1] * pm[1]) + xb[2] * pm[2]) + u * pm[3];
        xu[t * xstep + xoff] = (HELP && m_4 != 0.0) ? dsq : pk;
        WAVE_SYNC();
        __hip_atomic_store(prog, epoch + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 3; ++r) xb[r] = xa[r] + pb.kappa * (xb[r] * xb[r] * xb[r]);
    }
    xu[N * xstep + xoff] = (xb[0] * pm[0] + xb[1] * pm[1]) + xb[2] * pm[2];          // x_N (lanes j >= 3: 0)
    WAVE_SYNC();
    __hip_atomic_store(prog, epoch + N + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!HELP && MODE == 1 && l == 0) {
        st.d_c[b * st.E] = dnan ? NAN : sqrt(dmax);
        st.flag_c[b * st.E] = 0;
    }
}

// the steps first, first + stride, ... of the trajectory (and, first == 0, the terminal tile)
// NT: no tile records (the sweeps of the same workgroup are fly sweeps): only [x_t; u_t] and the [c_x | c_u | c] row are stored
// UL (deviation-form rollouts, rollacl_body): the record holds x_t only; this wave forms u_t = l_t + eps dl_t + L_t (x_t - xbar_t) itself
// (the reference's expression, :82, with rollrec_body's operands and summation order) and the squared step norms of d (:517-519).
// pool: an LDS counter the linearising waves draw their next step from (whoever is free takes it; null: first, first + stride, ...);
// term: 1 this wave also writes x_N and the terminal tile, 0 it does not (-1: the wave with first == 0).
template <int MODE, bool CTV, bool HELP, bool NT = false, bool UL = false>
__device__ __forceinline__ void rolllin_body(const RolloutArgs &a, const int b, double *const shxu, const double *const xu, int *const prog, const int epoch,
                                             const int first, const int stride, unsigned long long *const d_acc,
                                             const double *const stg = nullptr, const double eps = 0.0, int *const pool = nullptr,
                                             const int term = -1) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l = lane_, j = l & 15, g = l >> 4;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    int nom, lsel;
    double eps_in;
    if (!rollout_active<MODE>(st, b, nom, lsel, eps_in)) return;
    const int slot_n = b * (st.E + 1) + nom;
    const int slot_o = (MODE == 0) ? slot_n : cand_slot(b, 0, nom, st.E);
    double *__restrict__ xo = st.xs + (long)slot_o * st.x_stride;
    double *__restrict__ uo = st.us + (long)slot_o * st.u_stride;
    double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot_o) * st.tile_stride;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    double cf[4] = {0, 0, 0, 0};
    double es[4];
    double cq0 = 0, cq1 = 0, cq2 = 0, cpr = 0, clin = 0, cq00 = 0;
    const double mq = (j < 12) ? 1.0 : 0.0;
    const int jx = (j < 12) ? j : 11, j3 = j & 3;
    const int pkc = 4 * j3 + g;
#pragma unroll
    for (int s = 0; s < 4; ++s) es[s] = (j == 4 * s + g) ? 1.0 : 0.0;
    const double dgz[3] = {es[0], es[1], es[2]};
    const double zt0 = pb.Zt[l], zt1 = pb.Zt[64 + l], zt2 = pb.Zt[128 + l];
    if (!CTV) {
#pragma unroll
        for (int s = 0; s < 4; ++s) cf[s] = pb.Ctab[64 * s + l];
        cq0 = cf[0] * mq; cq1 = cf[1] * mq; cq2 = cf[2] * mq; cpr = cf[3];
        clin = pb.lin[pkc];
        cq00 = pb.q0[0];
    }
    double *const pxu = (j < 3) ? xo + 4 * j + g : (j == 3 ? uo + g : tile0 + TS_PAD);
    const long sxu = (j < 3) ? XSTR : (j == 3 ? USTR : TSTRIDE);
    const int qoff = (j < 4) ? TS_QR + 4 * j + g : (l == 4 ? TS_q : TS_PAD);
    const int c34 = TS_REG(3, l), r5 = TS_REG(5, l);
    const double pm[4] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0, j == 3 ? 1.0 : 0.0};
    const double m_j4 = (j < 4) ? 1.0 : 0.0, m_l4 = (l == 4) ? 1.0 : 0.0;
    const int pkoff = (j < 4) ? 4 * j + g : 0;
    auto wait_for = [&](const int want) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - want < 0) __builtin_amdgcn_s_sleep(1);
        WAVE_SYNC();
    };
    double dmax = 0.0;
    bool dnan = false;
    constexpr int cL_ = STG_CL, cX_ = STG_CX, cU_ = STG_CU;
    const double *const sL = stg, *const sX = stg + cL_ * 64, *const sl = sX + cX_ * 64, *const sdl = sl + cU_ * 64;
    const double m_j3 = (j < 3) ? 1.0 : 0.0;
    const int pkoff3 = (j < 3) ? 4 * j + g : 0;
    int have = 0, nextt = 0;                            // steps are drawn from the pool two at a time (one LDS atomic round trip per pair)
    auto draw = [&]() -> int {
        if (!have) {
            int v = 0;
            if (l == 0) v = __hip_atomic_fetch_add(pool, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            nextt = __builtin_amdgcn_readfirstlane(v);
            have = 2;
        }
        --have;
        return nextt++;
    };
    [[maybe_unused]] int nsteps = 0;
    for (int t = pool ? draw() : first; t < N; t = pool ? draw() : t + stride) {
        ++nsteps;
        wait_for(epoch + t + 1);
        const double *xt = xu + t * XU_REC;
        if (!UL && HELP && MODE == 1) {                             // d = maximum(norm(l_t - u_t))  (:517-519): same sum as rollin_body
            const double dn2 = ((xt[16] + xt[17]) + xt[18]) + xt[19];
            dnan |= (dn2 != dn2);
            dmax = (dn2 > dmax) ? dn2 : dmax;
        }
        const double xb[3] = {xt[g], xt[4 + g], xt[8 + g]};
        double u, pk;
        if (UL) {
            const double c_l = sl[t * USTR + g], c_dl = sdl[t * USTR + g];
            d4 fb = MFMA(sL[t * LSTR + j3 * 12 + g], xb[0] - sX[t * XSTR + g], zero4);               // L_t (x_t - xbar_t)   (:82)
            fb = MFMA(sL[t * LSTR + j3 * 12 + 4 + g], xb[1] - sX[t * XSTR + 4 + g], fb);
            fb = MFMA(sL[t * LSTR + j3 * 12 + 8 + g], xb[2] - sX[t * XSTR + 8 + g], fb);
            const double lnew = c_l + eps * c_dl;                   // l + eps dl           (:509)
            u = lnew + fb[0];
            const double du = c_l - u;                              // d = maximum(norm(l_t - u_t))  (:517-519), as rollin_body
            const double dsq = du * du;
            const double dn2 = ((readlane_f64(dsq, 0) + readlane_f64(dsq, 16)) + readlane_f64(dsq, 32)) + readlane_f64(dsq, 48);
            dnan |= (dn2 != dn2);
            dmax = (dn2 > dmax) ? dn2 : dmax;
            pk = xt[pkoff3] * m_j3 + u * pm[3];
        } else {
            u = xt[12 + g];
            pk = xt[pkoff] * m_j4;
        }
        double *__restrict__ tp = tile0 + (long)t * TSTRIDE;
        if (CTV) {
            const double *__restrict__ C = pb.Ctab + (long)t * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) cf[s] = C[64 * s + l];
            cq0 = cf[0] * mq; cq1 = cf[1] * mq; cq2 = cf[2] * mq; cpr = cf[3];
            clin = pb.lin[(long)t * 16 + pkc];
            cq00 = pb.q0[t];
        }
        pxu[(long)t * sxu] = pk;
        if (!NT) {
            const double z0 = fx_diag(zt0, dgz[0], pb.kappa, xb[0]);
            const double z1 = fx_diag(zt1, dgz[1], pb.kappa, xb[1]);
            const double z2 = fx_diag(zt2, dgz[2], pb.kappa, xb[2]);
            double2 *__restrict__ t2 = reinterpret_cast<double2 *>(tp);
            t2[l] = make_double2(z0, z1);
            t2[64 + l] = make_double2(z2, cpr);
            *reinterpret_cast<double2 *>(tp + c34) = make_double2(cq0, cq1);
            tp[r5] = cq2;
        }
        d4 cx = MFMA(cf[0], xb[0], zero4);                      // C [x;u] in B-form
        cx = MFMA(cf[1], xb[1], cx);
        cx = MFMA(cf[2], xb[2], cx);
        cx = MFMA(cf[3], u, cx);
        const double acc = ((cx[0] * pm[0] + cx[1] * pm[1]) + cx[2] * pm[2]) + cx[3] * pm[3];
        const double w = row_sum16(cost_term(pk, acc, clin));
        const double part = ((readlane_f64(w, 0) + readlane_f64(w, 16)) + readlane_f64(w, 32)) + readlane_f64(w, 48);
        tp[qoff] = fma(m_l4, part + cq00, m_j4 * (acc + clin));
    }
    if (UL) ACL_MARK(a.dump, (threadIdx.x >> 6), 3, nsteps);
    if ((HELP || UL) && MODE == 1 && l == 0) {
        atomicMax(&d_acc[0], (unsigned long long)__double_as_longlong(dmax));     // doubles >= +0 order like their bit patterns
        if (dnan) atomicOr(&d_acc[1], 1ull);
    }
    if (term < 0 ? (first != 0) : (term == 0)) return;
    // x_N and the terminal tile: h, h_x, h_xx at x_N   (ileqg.jl:314-316), as rollin_body
    wait_for(epoch + N + 1);
    {
        const double x = (l < 12) ? xu[N * XU_REC + l] : 0.0;
        if (l < 12) { xo[(long)N * XSTR + l] = x; shxu[l] = x; }
        WAVE_SYNC();
        double *__restrict__ tp = tile0 + (long)N * TSTRIDE;
        for (int e = l; e < 144; e += 64) tp[TT_Q + e] = pb.Qf[e];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc = fma(pb.Qf[jx * 12 + q], shxu[q], acc);
        const double qvf = pb.qvf[jx];
        if (l < 12) tp[TT_QV + l] = acc + qvf;
        const double part = row_sum16((j < 12) ? cost_term(shxu[jx], acc, qvf) : 0.0);
        if (l == 0) tp[TT_q] = part + pb.q0f;
        WAVE_SYNC();
    }
}

// =====================================================================================================
// Closed-loop rollouts in DEVIATION FORM (round 4; solve_block_kernel's split geometry, LQ family): the latency-bound shards
// (1024 / N samples per GPU, a sample owns 2..4 wavefronts) spent 100 k of their 650 k cycles in two closed-loop rollouts whose
// recursion wave carried 7 dependent-or-queued f64 MFMAs per step (3 for L_t (x_t - xbar_t), 1 for B u_t, 3 for A x_t: ~1000 cycles).
// With dx_t = x_t - xbar_t and the nominal pair (xbar, l) a trajectory of the same dynamics (xbar_{t+1} = f(xbar_t, l_t)),
//     x_{t+1} = A x_t + B (l_t + eps dl_t + L_t dx_t) + kappa x_t^3                                            (ileqg.jl:82-83, f of the LQ family)
//  => dx_{t+1} = (A + B L_t) dx_t + eps B dl_t + kappa (x_t^3 - xbar_t^3),          x_{t+1} = xbar_{t+1} + dx_{t+1}
// so the recursion wave issues THREE MFMAs per step (Acl_t dx_t, accumulator started at eps B dl_t) and nothing else on its chain:
// ~260 cycles per step.  Acl_t' = A' + L_t' B' and B dl_t are ONE MFMA per step (rows 0..11 and row 12 of the same product, the natural
// [L_t | dl_t] rows being its A operand), formed by a PRODUCER wave a few steps ahead of the recursion through an LDS ring; the
// controls u_t = l_t + eps dl_t + L_t dx_t (the reference's own expression, from the posted x_t), d = max ||l_t - u_t||, the cost
// gradient rows and the history stores follow BEHIND the recursion on the linearising waves, as before.
// Arithmetic: the same trajectory up to rounding (the feedback correction is formed on dx as in the reference; what differs is the order
// of the sums: ~1e-16 relative per step), so this geometry is NOT bit-identical to the one-wave rollouts of the other paths -- values
// agree to ~1e-13, iteration / line-search counts are equal (tests/test_gpu_block.py) and parity against the oracle is unchanged
// (1e-9).  OPT-IN (FusedArgs.acl, switch block_acl = 1): measured on MI355X the phase went 50 k -> 37 k cycles at 128 samples per GPU (batch
// 0.318 ms either way: the recursion wave reaches ~600 cycles per step, not 260 -- three DEPENDENT f64 MFMAs take ~300, the post /
// poll / loop overhead of the 256-register kernel another 250 -- and the linearising waves ~1,700 per step) and is slower at 512
// samples, where two waves share the linearisation; the default stays the round-2/3 split rollouts, bit-identical to every other path.
// =====================================================================================================
#define ACL_RING 8                                   /* steps the producer may run ahead of the recursion */

#define ACL_REC 160                                  /* doubles per ring slot: Acl' rows 0..11 on their 48 live lanes x 3 registers (144), then B dl_t [16] */
#define ACL_DOUBLES (ACL_RING * ACL_REC)

// every wave of the workgroup copies its share of the trajectory's operands (L, xbar, l, dl: rollin_body's STAGE layout) into LDS.
// Whole 64-double chunks, unclamped: the pools carry STG_PAD doubles of slack behind their last slot (alloc_state), the part of a chunk
// beyond the trajectory's own data is never read back, and every load is base + immediate offset (per-lane clamped addresses, one
// register pair per chunk, spilled to scratch in this 256-register kernel and serialised the loads: 28 k cycles instead of 2 k).
template <int NP>
__device__ __forceinline__ void stage_shared(const RolloutArgs &a, const int b, const int nom, const int lsel, double *const stg, const int part) {
    const int l = threadIdx.x & 63;
    const StateDev &st = a.st;
    const int N = st.N;
    const int slot_n = b * (st.E + 1) + nom;
    const double *__restrict__ xbar = st.xs + (long)slot_n * st.x_stride + l;
    const double *__restrict__ lnom = st.us + (long)slot_n * st.u_stride + l;
    const double *__restrict__ Lb = st.L + (long)lsel * st.l_half + (long)b * N * LSTR + l;
    const double *__restrict__ dlb = st.dl + (long)lsel * st.dl_half + (long)b * N * USTR + l;
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    double *const sL = stg + l, *const sX = sL + cL * 64, *const sl = sX + cX * 64, *const sdl = sl + cU * 64;
    // the gain rows are 2/3 of the bytes: dealt over all NP waves; the last wave also takes xbar, l, dl
    constexpr int LPER = (cL + NP - 1) / NP;
    double tL[LPER], tX[cX], tl[cU], tdl[cU];
    const double *__restrict__ Lw = Lb + (long)part * LPER * 64;
    const int nl = (cL - part * LPER < LPER) ? cL - part * LPER : LPER;          // (wave-uniform)
#pragma unroll
    for (int q = 0; q < LPER; ++q) tL[q] = xld(&Lw[64 * q]);
    if (part == NP - 1) {
#pragma unroll
        for (int q = 0; q < cX; ++q) tX[q] = xbar[64 * q];
#pragma unroll
        for (int q = 0; q < cU; ++q) { tl[q] = lnom[64 * q]; tdl[q] = xld(&dlb[64 * q]); }
    }
    double *const sLw = sL + part * LPER * 64;
#pragma unroll
    for (int q = 0; q < LPER; ++q) if (q < nl) sLw[64 * q] = tL[q];
    if (part == NP - 1) {
#pragma unroll
        for (int q = 0; q < cX; ++q) sX[64 * q] = tX[q];
#pragma unroll
        for (int q = 0; q < cU; ++q) { sl[64 * q] = tl[q]; sdl[64 * q] = tdl[q]; }
    }
}

__device__ __forceinline__ void spin_until(int *const word, const int want) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - want < 0) { }
    WAVE_SYNC();
}

// PRODUCER: Acl_t' = A' + L_t' B' (rows 0..11, register r on lane (g, j): the A operand of K-slice r of Acl_t (.)) and B dl_t (row 12)
// of step t = 0 .. N-1 into ring slot t mod ACL_RING, at most ACL_RING steps ahead of the recursion (its progress word `prog`).
__device__ __forceinline__ void rollprod_body(const RolloutArgs &a, const double *const stg, double *const ring, int *const pprog, int *const prog, const int epoch) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l = lane_, j = l & 15, g = l >> 4;
    const ProblemDev &pb = a.pb;
    const int N = a.st.N;
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    const double *const sL = stg, *const sdl = stg + (cL + cX + cU) * 64;
    const double mq = (j < 12) ? 1.0 : 0.0;
    const int jx = (j < 12) ? j : 11;
    d4 cA;                                              // A' in the accumulator layout = the A operand slices of [A | B] (rollrec_body's zA)
#pragma unroll
    for (int s = 0; s < 3; ++s) cA[s] = pb.Zt[jx * 16 + 4 * s + g] * mq;
    cA[3] = 0.0;
    const double bop = pb.Zt[jx * 16 + 12 + g] * mq;    // B' as the B operand: lane (g, j) = B[j][g]
    // natural [L_t | dl_t] rows: lane (g, j) = L_t[g][j] (j < 12), dl_t[g] (j == 12), 0 beyond
    const int loff = (j < 12) ? g * 12 + j : 0, doff = g;
    const double mL = (j < 12) ? 1.0 : 0.0, mD = (j == 12) ? 1.0 : 0.0;
    // live lanes of the three registers; dead lanes (columns 12..15: zeros) all write the slot's one zero position
    const int woff0 = (j < 12) ? 12 * g + j : 144 + 12, woff1 = (j < 12) ? 48 + 12 * g + j : 144 + 12, woff2 = (j < 12) ? 96 + 12 * g + j : 144 + 12;
    // next step's gain rows are requested before this step's MFMA; the recursion's progress word is only polled when the last value seen
    // does not yet free the ring slot (it runs at most ACL_RING steps behind)
    int seen = epoch;
    double nL = sL[loff], nD = sdl[doff];
    for (int t = 0; t < N; ++t) {
        const double la = nL * mL + nD * mD;
        const int tn = (t + 1 < N) ? t + 1 : t;
        nL = sL[tn * LSTR + loff]; nD = sdl[tn * USTR + doff];
        const d4 acl = MFMA(la, bop, cA);
        if (t >= ACL_RING && seen - (epoch + (t - ACL_RING) + 1) < 0) {            // the recursion has taken step t - ACL_RING's operands
            do { seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); } while (seen - (epoch + (t - ACL_RING) + 1) < 0);
            WAVE_SYNC();
        }
        double *const slot = ring + (t % ACL_RING) * ACL_REC;
        slot[woff0] = acl[0] * mq;
        slot[woff1] = acl[1] * mq;
        slot[woff2] = acl[2] * mq;
        if (g == 0) slot[144 + j] = acl[3] * mq;        // row 12 of the product: (B dl_t)[j]; columns 12..15 are zeros
        WAVE_SYNC();
        __hip_atomic_store(pprog, epoch + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// RECURSION: dx_{t+1} = Acl_t dx_t + eps B dl_t + kappa (x_t^3 - xbar_t^3); posts x_t = xbar_t + dx_t (packed) and the progress word.
__device__ __forceinline__ void rollacl_body(const RolloutArgs &a, const double eps, const double *const stg, const double *const ring, double *const xu,
                                             int *const pprog, int *const prog, const int epoch) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l = lane_, j = l & 15, g = l >> 4;
    const int N = a.st.N;
    const double kappa = a.pb.kappa;
    const bool has_cubic = kappa != 0.0;
    constexpr int cL = STG_CL;
    const double *const sX = stg + cL * 64;
    const double pm[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
    const int xoff = (j < 3) ? 4 * j + g : (ROLLIN_NST + 1) * XU_REC + l;         // packed x: lane (g, s), s < 3, holds component 4 s + g; idle lanes: dump
    const int xstep = (j < 3) ? XU_REC : 0;
    const int roff0 = (j < 12) ? 12 * g + j : 144 + 12, roff1 = (j < 12) ? 48 + 12 * g + j : 144 + 12, roff2 = (j < 12) ? 96 + 12 * g + j : 144 + 12;
    double dx[3] = {0.0, 0.0, 0.0};                                               // x_0 = xbar_0 (:73)
    double xb[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) xb[r] = sX[4 * r + g];
    // The step's operands are requested one step ahead, under the MFMAs of the step before, and the producer's progress word is only
    // polled when the last value seen does not cover the step wanted (the producer runs up to ACL_RING steps ahead: one poll per few steps).
    int known = epoch;
    auto ensure = [&](const int want) {
        if (known - want < 0) {
            do { known = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); } while (known - want < 0);
            WAVE_SYNC();
        }
    };
    // One step: `cur` holds its operands (Acl_t' slices, B dl_t), `nxt` receives step t + 1's from ring slot KN (a compile-time
    // position: the time loop is unrolled over the ring, so no LDS address is computed per step and the two operand sets alternate
    // without copies).  The next operands are REQUESTED before this step's MFMAs are issued: their LDS round trips run under the
    // matrix pipe (left to the scheduler the loads sank to the top of the next iteration, each in front of its consumer: 880 cycles per step).
    struct Ops { double a0, a1, a2, p0, p1, p2; };
    const double *const r0 = ring + roff0, *const r1 = ring + roff1, *const r2 = ring + roff2, *const rp = ring + 144 + g;
    double *const xup = xu + xoff;
#ifdef RAT_DIAG_PHASES
    unsigned long long dgs[4] = {0, 0, 0, 0}, dgt = __builtin_readcyclecounter();
#define CH_STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); dgs[i_] += n_ - dgt; dgt = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CH_STAMP(i_) do {} while (0)
#endif
    auto step = [&](const int t, const int KN, const Ops &cur, Ops &nxt) {
        double x[3], cub[3], xn[3];
        CH_STAMP(3);
#pragma unroll
        for (int r = 0; r < 3; ++r) x[r] = xb[r] + dx[r];
        xup[t * xstep] = (x[0] * pm[0] + x[1] * pm[1]) + x[2] * pm[2];          // x_t = xbar_t + dx_t to the linearising waves
        WAVE_SYNC();                                  // (the post and the slot's loads are issued before the word that announces them)
        __hip_atomic_store(prog, epoch + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 3; ++r) cub[r] = 0.0;
        if (has_cubic) {                               // (wave-uniform: a linear f skips the 18 vector instructions of the drift term)
#pragma unroll
            for (int r = 0; r < 3; ++r) cub[r] = kappa * (x[r] * x[r] * x[r]) - kappa * (xb[r] * xb[r] * xb[r]);
        }
        d4 acc = {eps * cur.p0, eps * cur.p1, eps * cur.p2, 0.0};
        CH_STAMP(0);
        if (t + 1 < N) ensure(epoch + t + 2);
        nxt.a0 = r0[KN * ACL_REC]; nxt.a1 = r1[KN * ACL_REC]; nxt.a2 = r2[KN * ACL_REC];
        nxt.p0 = rp[KN * ACL_REC]; nxt.p1 = rp[KN * ACL_REC + 4]; nxt.p2 = rp[KN * ACL_REC + 8];
#pragma unroll
        for (int r = 0; r < 3; ++r) xn[r] = sX[(t + 1) * XSTR + 4 * r + g];
        CH_STAMP(1);
        acc = MFMA(cur.a0, dx[0], acc);
        acc = MFMA(cur.a1, dx[1], acc);
        acc = MFMA(cur.a2, dx[2], acc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r) { dx[r] = acc[r] + cub[r]; xb[r] = xn[r]; }
        asm volatile("" :: "v"(dx[0]), "v"(dx[1]), "v"(dx[2]));
        CH_STAMP(2);
    };
    ensure(epoch + 1);
    Ops oa, ob;
    oa.a0 = r0[0]; oa.a1 = r1[0]; oa.a2 = r2[0]; oa.p0 = rp[0]; oa.p1 = rp[4]; oa.p2 = rp[8];
    static_assert(ACL_RING % 2 == 0, "the operand sets alternate over an even ring");
    for (int t0 = 0; t0 < N; t0 += ACL_RING) {
#pragma unroll
        for (int k = 0; k < ACL_RING; k += 2) {
            if (t0 + k < N) step(t0 + k, (k + 1) % ACL_RING, oa, ob);
            if (t0 + k + 1 < N) step(t0 + k + 1, (k + 2) % ACL_RING, ob, oa);
        }
    }
#ifdef RAT_DIAG_PHASES
    for (int q = 0; q < 4; ++q) ACL_MARK(a.dump, 0, 4 + q, dgs[q]);
#endif
    xu[N * xstep + xoff] = ((xb[0] + dx[0]) * pm[0] + (xb[1] + dx[1]) * pm[1]) + (xb[2] + dx[2]) * pm[2];      // x_N
    WAVE_SYNC();
    __hip_atomic_store(prog, epoch + N + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// =====================================================================================================
// rollprl_body: the closed-loop rollout of ONE candidate TIME-PARALLEL over the four wavefronts of its workgroup (solve_block_psw_kernel;
// LQ family with kappa == 0, time-invariant cost, operands staged in LDS by stage_shared; switch psw_prl).  simulate_dynamics (ileqg.jl:62-87)
// is a chain of N dependent steps that ONE wavefront walks (rollacl_body) while three others linearise behind it; but for kappa == 0 the
// DEVIATION dx_t = x_t - xbar_t obeys an affine recursion, dx_{t+1} = (A + B L_t) dx_t + eps B dl_t (xbar is a trajectory of the same
// dynamics), and affine maps compose.  The horizon is cut into four segments [cut_w, cut_w+1):
//   wave 0          runs the ordinary rollout (recursion + linearisation, rollin_body's step) over segment 0 from x_0 at once;
//   wave w >= 1     first builds the ELEMENT of segment w - 1 -- the map dx_start -> dx_end = Phi dx_start + c -- by running the deviation
//                   recursion on THIRTEEN columns at once: in B-form a vector is replicated over the 16 columns of the MFMA's B operand and
//                   every column is computed on its own, so columns 0..11 carry the unit vectors (they become Phi) and column 12 the affine
//                   input (it becomes c): the recursion's own seven MFMAs per step, no linearisation, ~half an ordinary step;
//                   takes the deviation at the start of segment w - 1 from wave w - 1 (zero for w = 1), "hops" to its own start
//                   (dx by column through three MFMAs, then Phi dx + c by a row sum: ~400 cycles), hands that on, and runs the ordinary
//                   rollout over segment w from x = xbar + dx.
// Every x_t, u_t, cost row and step norm comes from the ordinary step's own arithmetic on a start state that differs from the sequential
// one by rounding (the hop): results agree with the sequential rollout to ~1e-16, like everything else on this path.  Critical path: with
// the cuts balanced (rollprl_cuts) ~0.35 N ordinary steps instead of N recursion steps followed by the linearising pool's tail.
// =====================================================================================================
struct PrlShared { double box[PRL_WAVES][192]; int flag[PRL_WAVES]; };

__device__ __forceinline__ void rollprl_body(const RolloutArgs &a, const int b, const int nom, const double eps, const double *const stg, double *const shxu,
                                             PrlShared *const ps, const int epoch, const int wave, const PrlCuts &pc, unsigned long long *const d_acc) {
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int l = lane_, j = l & 15, g = l >> 4;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    const int slot_o = cand_slot(b, 0, nom, st.E);
    double *__restrict__ xo = st.xs + (long)slot_o * st.x_stride;
    double *__restrict__ uo = st.us + (long)slot_o * st.u_stride;
    double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot_o) * st.tile_stride;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    const double *const sL = stg, *const sX = stg + cL * 64, *const sl = sX + cX * 64, *const sdl = sl + cU * 64;
    // per-lane constants of rollin_body (LQ family, time-invariant cost)
    double zA[4], cf[4], es[4], lin4[4];
    const double mq = (j < 12) ? 1.0 : 0.0;
    const int jx = (j < 12) ? j : 11, j3 = j & 3;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        es[s] = (j == 4 * s + g) ? 1.0 : 0.0;
        zA[s] = pb.Zt[jx * 16 + 4 * s + g] * mq;
        cf[s] = pb.Ctab[64 * s + l];
        lin4[s] = pb.lin[4 * s + g];
    }
    const double cq00 = pb.q0[0];
    const double pm[4] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0, j == 3 ? 1.0 : 0.0};
    double *const pxu = (j < 3) ? xo + 4 * j + g : (j == 3 ? uo + g : tile0 + TS_PAD);
    const long sxu = (j < 3) ? XSTR : (j == 3 ? USTR : TSTRIDE);
    const int t_lo = pc.cut[wave], t_hi = pc.cut[wave + 1];
#ifdef RAT_DIAG_PHASES
#define PRL_STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0 && b < 8 && a.dump) \
        a.dump[3072 + b * 64 + wave * 16 + (i_)] = (double)__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PRL_STAMP(i_) do {} while (0)
#endif
    PRL_STAMP(0);
    // the terminal tile's constants are requested now (their L2 round trips pass under the segment): wave 0 copies Qf after its segment,
    // the last wave holds its row of Qf for Qf x_N
    double *__restrict__ tp = tile0 + (long)N * TSTRIDE;
    double qfc[3] = {0.0, 0.0, 0.0}, qfr[12], qvf = 0.0;
#pragma unroll
    for (int q = 0; q < 12; ++q) qfr[q] = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) qfc[q] = pb.Qf[(l + 64 * q < 144) ? l + 64 * q : 0];
    } else if (wave == PRL_WAVES - 1) {
#pragma unroll
        for (int q = 0; q < 12; ++q) qfr[q] = pb.Qf[jx * 12 + q];
        qvf = pb.qvf[jx];
    }
    // ---- waves 1..3: the element of the segment before their own, then the hop ----------------------------------------------------------
    double dx[3] = {0.0, 0.0, 0.0};                             // deviation at this wave's first step, B-form
    if (wave >= 1) {
        double D[3] = {es[0], es[1], es[2]};                    // columns 0..11: unit vectors; column 12: zero (the affine part)
        const double m12 = (j == 12) ? eps : 0.0;
        // Five MFMAs per step, three of them in the chain: Acl_t' = A' + L_t' B' in the accumulator layout IS the A operand of Acl_t (.)
        // slice by slice (rollprod_body's product), and the affine column's input B (eps dl_t) starts the accumulator -- both independent of D
        // (an f64 MFMA holds the datapath for 64 cycles whatever it computes: the count is the cost, profiles/r01_ubench_fp64_pipe.md).
        d4 cA;
#pragma unroll
        for (int s = 0; s < 3; ++s) cA[s] = zA[s];              // (zA[s] on lane (g, j) = A[j][4 s + g] = A'[4 s + g][j])
        cA[3] = 0.0;
        const double bop = zA[3];                               // B' as the B operand: lane (g, j) = B[j][g]
        const int loff = g * 12 + jx;
        for (int t = pc.cut[wave - 1]; t < t_lo; ++t) {
            const double la = sL[t * LSTR + loff] * mq;         // natural rows of L_t: lane (g, j) = L_t[g][j]
            const double dlt = sdl[t * USTR + g];
            const d4 acl = MFMA(la, bop, cA);
            d4 xa = MFMA(zA[3], m12 * dlt, zero4);
            xa = MFMA(acl[0], D[0], xa);
            xa = MFMA(acl[1], D[1], xa);
            xa = MFMA(acl[2], D[2], xa);
            D[0] = xa[0]; D[1] = xa[1]; D[2] = xa[2];
        }
        PRL_STAMP(1);
        double xin[3] = {0.0, 0.0, 0.0};
        if (wave >= 2) {                                        // the deviation at the start of that segment, from the wave before
            // (bounded: the producer is a wave of this workgroup that cannot fail to post; a protocol error must not hang the device)
            for (int polls = 0; polls < (1 << 24) &&
                 __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ps->flag[wave - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - (epoch + 1) < 0; ++polls)
                __builtin_amdgcn_s_sleep(1);
            WAVE_SYNC();
#pragma unroll
            for (int s = 0; s < 3; ++s) xin[s] = ps->box[wave - 1][64 * s + l];
        }
        PRL_STAMP(2);
        // hop: dx_out = Phi dx_in + c.  dx_in by column (lane (., j) <- component j): three MFMAs against the unit slices; Phi's column j times
        // it, the affine column added, summed over the row's sixteen lanes: component 4 s + g on every lane of row g -- B-form again
        d4 tj = MFMA(xin[0], es[0], zero4);
        tj = MFMA(xin[1], es[1], tj);
        tj = MFMA(xin[2], es[2], tj);
        const double c12 = (j == 12) ? 1.0 : 0.0;
#pragma unroll
        for (int s = 0; s < 3; ++s) dx[s] = row_sum16(fma(D[s], tj[0], D[s] * c12));
        if (wave < PRL_WAVES - 1) {
#pragma unroll
            for (int s = 0; s < 3; ++s) ps->box[wave][64 * s + l] = dx[s];
            WAVE_SYNC();
            if (l == 0) __hip_atomic_store(&ps->flag[wave], epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    PRL_STAMP(3);
    // ---- the ordinary rollout over [t_lo, t_hi) (rollin_body's closed-loop step: staged operands, no tile records, cost rows of five steps per
    //      set of four MFMAs) from x = xbar + dx ------------------------------------------------------------------------------------------------
    double xb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) xb[s] = sX[t_lo * XSTR + 4 * s + g] + dx[s];
    double xq[4] = {0.0, 0.0, 0.0, 0.0};
    double dmax = 0.0;
    bool dnan = false;
    constexpr int RD = 5;
    auto step = [&](const int t, const int d, const bool last) {
        const double c_l = sl[t * USTR + g], c_dl = sdl[t * USTR + g];
        const double xb0 = sX[t * XSTR + g], xb1 = sX[t * XSTR + 4 + g], xb2 = sX[t * XSTR + 8 + g];
        const double La0 = sL[t * LSTR + j3 * 12 + g], La1 = sL[t * LSTR + j3 * 12 + 4 + g], La2 = sL[t * LSTR + j3 * 12 + 8 + g];
        d4 xa = MFMA(zA[0], xb[0], zero4);
        xa = MFMA(zA[1], xb[1], xa);
        xa = MFMA(zA[2], xb[2], xa);
        d4 fb = MFMA(La0, xb[0] - xb0, zero4);                  // L_t (x_t - xbar_t)   (:82)
        fb = MFMA(La1, xb[1] - xb1, fb);
        fb = MFMA(La2, xb[2] - xb2, fb);
        const double lnew = c_l + eps * c_dl;                   // l + eps dl           (:509)
        const double u = lnew + fb[0];
        const double du = c_l - u, dsq = du * du;
        const double dn2 = ((readlane_f64(dsq, 0) + readlane_f64(dsq, 16)) + readlane_f64(dsq, 32)) + readlane_f64(dsq, 48);
        dnan |= (dn2 != dn2);
        dmax = (dn2 > dmax) ? dn2 : dmax;
        xa = MFMA(zA[3], u, xa);
        const double pk = ((xb[0] * pm[0] + xb[1] * pm[1]) + xb[2] * pm[2]) + u * pm[3];
        pxu[(long)t * sxu] = pk;
        const bool me = (j == d) || (last && j > d);
#pragma unroll
        for (int s = 0; s < 3; ++s) xq[s] = me ? xb[s] : xq[s];
        xq[3] = me ? u : xq[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) xb[r] = xa[r];              // (kappa == 0: no cubic term)
    };
    auto flush = [&](const int t0, const int cnt) {
        d4 cx = MFMA(cf[0], xq[0], zero4);                      // C [x;u] of every step of the group
        cx = MFMA(cf[1], xq[1], cx);
        cx = MFMA(cf[2], xq[2], cx);
        cx = MFMA(cf[3], xq[3], cx);
        double *__restrict__ rp = tile0 + (long)(t0 + ((j < cnt) ? j : cnt - 1)) * TSTRIDE;
#pragma unroll
        for (int s = 0; s < 4; ++s) rp[TS_QR + 4 * s + g] = cx[s] + lin4[s];                    // [c_x | c_u] = C [x;u] + [qv;rv]  (:297,:299)
        const double a0 = cost_term(xq[0], cx[0], lin4[0]), a1 = cost_term(xq[1], cx[1], lin4[1]);
        const double a2 = cost_term(xq[2], cx[2], lin4[2]), a3 = cost_term(xq[3], cx[3], lin4[3]);
        double wr[4];
        rows_bcast((a0 + a2) + (a1 + a3), wr);
        const double part = ((wr[0] + wr[1]) + wr[2]) + wr[3];
        rp[(g == 0) ? TS_q : TS_PAD + (g & 1)] = (g == 0) ? part + cq00 : 0.0;
    };
    int t0 = t_lo;
    for (; t0 + RD <= t_hi; t0 += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) step(t0 + d, d, d == RD - 1);
        flush(t0, RD);
    }
    {
        const int nt = t_hi - t0;
#pragma unroll
        for (int d = 0; d < RD - 1; ++d)
            if (d < nt) step(t0 + d, d, d == nt - 1);
        if (nt > 0) flush(t0, nt);
    }
    PRL_STAMP(4);
    if (l == 0) {
        atomicMax(&d_acc[0], (unsigned long long)__double_as_longlong(dmax));     // doubles >= +0 order like their bit patterns
        if (dnan) atomicOr(&d_acc[1], 1ull);
    }
    // ---- wave 0: Qf into the terminal tile; the last wave: x_N and the rest of that tile (rollin_body's epilogue) --------------------------
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (l + 64 * q < 144) tp[TT_Q + l + 64 * q] = qfc[q];
    }
    if (wave == PRL_WAVES - 1) {
        d4 tj = MFMA(xb[0], es[0], zero4);
        tj = MFMA(xb[1], es[1], tj);
        tj = MFMA(xb[2], es[2], tj);
        const double x = tj[0];
        if (l < 12) { xo[(long)N * XSTR + l] = x; shxu[l] = x; }
        WAVE_SYNC();
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc = fma(qfr[q], shxu[q], acc);
        if (l < 12) tp[TT_QV + l] = acc + qvf;
        const double part = row_sum16((j < 12) ? cost_term(shxu[jx], acc, qvf) : 0.0);
        if (l == 0) tp[TT_q] = part + pb.q0f;
        WAVE_SYNC();
    }
    PRL_STAMP(5);
#undef PRL_STAMP
}

#if RAT_PART & PART_ROLL
template <int MODEL, int MODE, bool CTV, bool SEP>
__global__ __launch_bounds__(64) void rollin_kernel(RolloutArgs a) {
    __shared__ double shxu[16];
    constexpr bool ST0 = (MODE == 0) && (MODEL == 1);          // open loop, LQ family: the nominal controls staged in LDS (N <= ROLLIN_NST)
    __shared__ double stg0[ST0 ? STG_CU * 64 : 1];
    if (ST0 && a.st.N <= ROLLIN_NST) rollin_body<MODEL, MODE, CTV, ST0, SEP>(a, blockIdx.x, shxu, stg0);
    else rollin_body<MODEL, MODE, CTV, false, SEP>(a, blockIdx.x, shxu);
}

// The E line-search candidates of a sample as the waves of ONE workgroup (eight per workgroup; N <= ROLLIN_NST): they share the sample's operands
// (L, xbar, l, dl), so the staged variant of rollin_body applies -- every wave copies the operands of the whole trajectory into the
// workgroup's LDS area (identical values: benign overlap, each wave reads back what it wrote itself) and its time loop issues no
// global load.  The unstaged kernel spends 60 % of its cycles in s_waitcnt on those loads (tools/profile_rollin.sh).
template <int MODEL, bool CTV, bool NOTILE>
__global__ __launch_bounds__(512) void rollin_stage_kernel(RolloutArgs a) {
    const int nb = (a.st.E + 7) >> 3;                           // workgroups per sample: eight candidates each
    const int b = blockIdx.x / nb, k = (blockIdx.x - b * nb) * 8 + (threadIdx.x >> 6);
    if (k >= a.st.E) return;
    __shared__ double shxu[8][16];
    __shared__ double stg[STG_DOUBLES];
    rollin_body<MODEL, 1, CTV, true, false, ROLLIN_PREFETCH, NOTILE>(a, b * a.st.E + k, shxu[(threadIdx.x >> 6) & 7], stg);
}

// =====================================================================================================
// rollin_multi_kernel: the closed-loop rollouts of ALL E <= 16 line-search candidates of a sample in ONE wavefront (speculative path,
// LQ family, candidate records without tiles: NOTILE).  In rollin_body a vector in B-form is replicated across the 16 columns of the
// MFMA's B operand; here column j carries candidate j -- x_t^(j), u_t^(j) = l_t + eps_j dl_t + L_t (x_t^(j) - xbar_t) -- so the very same
// 11 MFMAs per step ([A|B][x;u]: 4, L dx: 3, C [x;u]: 4) advance 16 trajectories instead of one.  An MFMA forms every element of D from
// its own row of A and column of B in a fixed order, whatever the other columns hold, and every vector instruction below is the
// per-lane expression of rollin_body: each candidate's x, u, [c_x | c_u], c, d are the bits the one-candidate kernels produce (tested).
// The sample's operands (L, xbar, l, dl) are staged in LDS once for all candidates; what differs per column is eps_j and the slot the
// lane stores to.  Row sums over the 16 components of a candidate -- which the one-candidate kernels take over a packed vector with DPP
// rotations and v_readlane -- are taken in the same order with the rows brought together by lane swaps (rows_bcast).
// One wavefront per sample: 1024 waves instead of 8192 for BASELINE config 3.
// =====================================================================================================
template <bool CTV>
__global__ __launch_bounds__(64) void rollin_multi_kernel(RolloutArgs a) {
    __shared__ double stg[STG_DOUBLES];
    __shared__ double shx[16][16];                               // x_N of candidate i by component
    const int l = threadIdx.x & 63, j = l & 15, g = l >> 4;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N, E = st.E;
    const int b = blockIdx.x;
    const int v_act = st.ls_active[b], v_nom = st.slot_nom[b], v_lsel = st.lsel[b];
    const int s_act = wave_uniform(v_act), nom = wave_uniform(v_nom), lsel = wave_uniform(v_lsel);
    const double eps_in = st.ls_eps[b];
    if (!s_act) return;
    const bool live = j < E;                                    // this lane's column carries a candidate
    const int k = live ? j : E - 1;
    const int slot_n = b * (E + 1) + nom, slot_o = cand_slot(b, k, nom, E);
    const double *__restrict__ xbar = st.xs + (long)slot_n * st.x_stride;
    const double *__restrict__ lnom = st.us + (long)slot_n * st.u_stride;
    const double *__restrict__ Lb = st.L + (long)lsel * st.l_half + (long)b * N * LSTR;
    const double *__restrict__ dlb = st.dl + (long)lsel * st.dl_half + (long)b * N * USTR;
    double *const xo = st.xs + (long)slot_o * st.x_stride, *const uo = st.us + (long)slot_o * st.u_stride;
    double *const tile0 = st.tiles + tile_slot(st, b, slot_o) * st.tile_stride;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    double eps = eps_in;
    for (int q = 0; q < E; ++q) if (q < k) eps *= a.op.lambda;   // eps_k = eps * lambda^k by repeated multiplication (:530,:557)
    const double mq = (j < 12) ? 1.0 : 0.0;
    const int jx = (j < 12) ? j : 11, j3 = j & 3;
    double zA[4], cf[4] = {0, 0, 0, 0}, lin[4] = {0, 0, 0, 0}, es[3], cq00 = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) zA[s] = pb.Zt[jx * 16 + 4 * s + g] * mq;
#pragma unroll
    for (int s = 0; s < 3; ++s) es[s] = (j == 4 * s + g) ? 1.0 : 0.0;
    if (!CTV) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { cf[s] = pb.Ctab[64 * s + l]; lin[s] = pb.lin[4 * s + g]; }
        cq00 = pb.q0[0];
    }
    // per-lane store targets.  Columns j >= E carry a copy of candidate E - 1 (k is clamped): they compute the same values and store them
    // to the same addresses as column E - 1, so every store stays unconditional without a shared dump location (1024 waves writing one
    // cache line on every step serialise on its L2 channel: measured 80 us per launch with such a sink).
    double *const px = xo + g, *const pu = uo + g, *const pq = tile0 + TS_QR + g;
    constexpr long sx = XSTR, su = USTR, sq = TSTRIDE;
    constexpr int o4 = 4;
    double *const pc = tile0 + ((g == 0) ? TS_q : TS_PAD + (g & 1));      // row 0: c; rows 1..3: 0.0 to the record's zero pair
    constexpr long sc = TSTRIDE;
    // the sample's operands into LDS (rollin_body's STAGE layout)
    constexpr int cL = STG_CL, cX = STG_CX, cU = STG_CU;
    double *const sL = stg, *const sX = stg + cL * 64, *const sl = sX + cX * 64, *const sdl = sl + cU * 64;
    {
        double tL[cL], tX[cX], tl[cU], tdl[cU];
#pragma unroll
        for (int q = 0; q < cL; ++q) { const int e = 64 * q + l; tL[q] = Lb[(e < N * LSTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cX; ++q) { const int e = 64 * q + l; tX[q] = xbar[(e < (N + 1) * XSTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cU; ++q) { const int e = 64 * q + l; tl[q] = lnom[(e < N * USTR) ? e : 0]; tdl[q] = dlb[(e < N * USTR) ? e : 0]; }
#pragma unroll
        for (int q = 0; q < cL; ++q) sL[64 * q + l] = tL[q];
#pragma unroll
        for (int q = 0; q < cX; ++q) sX[64 * q + l] = tX[q];
#pragma unroll
        for (int q = 0; q < cU; ++q) { sl[64 * q + l] = tl[q]; sdl[64 * q + l] = tdl[q]; }
        WAVE_SYNC();
    }
    double xb[3];                                                // x_t of this column's candidate in B-form (all candidates start at xbar_0)
#pragma unroll
    for (int s = 0; s < 3; ++s) xb[s] = sX[4 * s + g];
    double dmax = -INFINITY;
    bool dnan = false;
    // operands of step t + 1 are read from LDS during step t (one wave per SIMD: nothing else hides the LDS latency)
    double n_l = sl[g], n_dl = sdl[g], n_xb[3], n_La[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { n_xb[s] = sX[4 * s + g]; n_La[s] = sL[j3 * 12 + 4 * s + g]; }
    for (int t = 0; t < N; ++t) {
        const double c_l = n_l, c_dl = n_dl;
        const double c_xb[3] = {n_xb[0], n_xb[1], n_xb[2]}, c_La[3] = {n_La[0], n_La[1], n_La[2]};
        {
            const int tn = (t + 1 < N) ? t + 1 : t;
            n_l = sl[tn * USTR + g]; n_dl = sdl[tn * USTR + g];
#pragma unroll
            for (int s = 0; s < 3; ++s) { n_xb[s] = sX[tn * XSTR + 4 * s + g]; n_La[s] = sL[tn * LSTR + j3 * 12 + 4 * s + g]; }
        }
        if (CTV) {
            const double *__restrict__ C = pb.Ctab + (long)t * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) { cf[s] = C[64 * s + l]; lin[s] = pb.lin[(long)t * 16 + 4 * s + g]; }
            cq00 = pb.q0[t];
        }
        d4 xa = MFMA(zA[0], xb[0], zero4);                        // x-part of [A|B][x; u]
        xa = MFMA(zA[1], xb[1], xa);
        xa = MFMA(zA[2], xb[2], xa);
        d4 fb = MFMA(c_La[0], xb[0] - c_xb[0], zero4);            // L_t (x_t - xbar_t)   (:82)
        fb = MFMA(c_La[1], xb[1] - c_xb[1], fb);
        fb = MFMA(c_La[2], xb[2] - c_xb[2], fb);
        const double lnew = c_l + eps * c_dl;                     // l + eps dl           (:509)
        const double u = lnew + fb[0];
        // d = maximum(norm(l_t - u_t))  (:517-519): squared norms, rooted once after the loop, rows added in rollin_body's order
        const double du = c_l - u, dsq = du * du;
        double dr[4];
        rows_bcast(dsq, dr);
        const double dn2 = ((dr[0] + dr[1]) + dr[2]) + dr[3];
        dnan |= (dn2 != dn2);
        dmax = (dn2 > dmax) ? dn2 : dmax;
        xa = MFMA(zA[3], u, xa);
        // [x_t; u_t] of every candidate
        px[(long)t * sx] = xb[0]; px[(long)t * sx + o4] = xb[1]; px[(long)t * sx + 2 * o4] = xb[2];
        pu[(long)t * su] = u;
        // [c_x | c_u] = C [x;u] + [qv; rv]  (:297,:299) and c (:296)
        d4 cx = MFMA(cf[0], xb[0], zero4);
        cx = MFMA(cf[1], xb[1], cx);
        cx = MFMA(cf[2], xb[2], cx);
        cx = MFMA(cf[3], u, cx);
#pragma unroll
        for (int s = 0; s < 4; ++s) pq[(long)t * sq + s * o4] = cx[s] + lin[s];
        // the packed vector of rollin_body holds component 4 s + g on lane (g, s): its row sum runs (a0 + a2) + (a1 + a3), rows in order
        const double a0 = cost_term(xb[0], cx[0], lin[0]), a1 = cost_term(xb[1], cx[1], lin[1]);
        const double a2 = cost_term(xb[2], cx[2], lin[2]), a3 = cost_term(u, cx[3], lin[3]);
        const double w = (a0 + a2) + (a1 + a3);
        double wr[4];
        rows_bcast(w, wr);
        const double part = ((wr[0] + wr[1]) + wr[2]) + wr[3];
        pc[(long)t * sc] = (g == 0) ? part + cq00 : 0.0;          // row 0: c; rows 1..3: exact zeros to the record's zero pair
#pragma unroll
        for (int r = 0; r < 3; ++r) xb[r] = xa[r] + pb.kappa * (xb[r] * xb[r] * xb[r]);
    }
    // x_N of every candidate by component: row i of D = sum_s xb[s]' es[s] is candidate i's state vector
    {
        d4 tj = MFMA(xb[0], es[0], zero4);
        tj = MFMA(xb[1], es[1], tj);
        tj = MFMA(xb[2], es[2], tj);
#pragma unroll
        for (int r = 0; r < 4; ++r) shx[4 * r + g][j] = tj[r];
        WAVE_SYNC();
    }
    // terminal tile of candidate 4 it + g on the lanes of row g: h, h_x, h_xx at x_N   (ileqg.jl:314-316), as rollin_body
    for (int it = 0; 4 * it < E; ++it) {
        const int i = 4 * it + g;
        const bool on = i < E;
        const int ii = on ? i : E - 1;
        const int so = cand_slot(b, ii, nom, E);
        double *__restrict__ tp = st.tiles + tile_slot(st, b, so) * st.tile_stride + (long)N * TSTRIDE;
        double *__restrict__ xoi = st.xs + (long)so * st.x_stride;
        const double x = (j < 12) ? shx[ii][j] : 0.0;
        if (on && j < 12) xoi[(long)N * XSTR + j] = x;
        if (on) for (int e = j; e < 144; e += 16) tp[TT_Q + e] = pb.Qf[e];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc = fma(pb.Qf[jx * 12 + q], shx[ii][q], acc);
        const double qvf = pb.qvf[jx];
        if (on && j < 12) tp[TT_QV + j] = acc + qvf;
        const double part = row_sum16((j < 12) ? cost_term(shx[ii][jx], acc, qvf) : 0.0);
        if (on && j == 0) tp[TT_q] = part + pb.q0f;
    }
    if (live && g == 0) {
        st.d_c[b * E + j] = dnan ? NAN : sqrt(dmax);
        st.flag_c[b * E + j] = 0;
    }
}

bool rollin_notile_supported(const ProblemDev &pb, const StateDev &st) { return pb.model == 1 && st.E > 1 && st.N <= ROLLIN_NST; }
void launch_rollin(const RolloutArgs &a, hipStream_t s) {
    const int ncand = (a.mode == 0) ? a.st.B : a.st.B * a.st.E;
    if (ncand <= 0) return;
    if (a.mode == 1 && a.notile && a.multi && a.pb.model == 1 && a.st.E > 1 && a.st.E <= 16 && a.st.N <= ROLLIN_NST) {
        if (a.pb.cost_tv) hipLaunchKernelGGL((rollin_multi_kernel<true>), dim3(a.st.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((rollin_multi_kernel<false>), dim3(a.st.B), dim3(64), 0, s, a);
        return;
    }
    if (a.mode == 1 && a.pb.model == 1 && a.st.E > 1 && a.st.N <= ROLLIN_NST) {
        const dim3 g2(a.st.B * ((a.st.E + 7) / 8)), b2(64 * (a.st.E < 8 ? a.st.E : 8));
        if (a.notile) {
            if (a.pb.cost_tv) hipLaunchKernelGGL((rollin_stage_kernel<1, true, true>), g2, b2, 0, s, a);
            else hipLaunchKernelGGL((rollin_stage_kernel<1, false, true>), g2, b2, 0, s, a);
        } else {
            if (a.pb.cost_tv) hipLaunchKernelGGL((rollin_stage_kernel<1, true, false>), g2, b2, 0, s, a);
            else hipLaunchKernelGGL((rollin_stage_kernel<1, false, false>), g2, b2, 0, s, a);
        }
        return;
    }
    const dim3 grid(ncand), block(64);
    const bool sep = (a.mode == 0) || a.st.E == 1;            // one wave per SIMD: see rollin_body
    // one compact instantiation per (model family, mode): the LQ hot loop must not carry the inlined pow() expansions
    // of the power-law family through the instruction cache
#define ROLLIN_LAUNCH(M, MD, C) do { if (sep) hipLaunchKernelGGL((rollin_kernel<M, MD, C, true>), grid, block, 0, s, a); \
                                     else hipLaunchKernelGGL((rollin_kernel<M, MD, C, false>), grid, block, 0, s, a); } while (0)
    if (a.pb.model == 1 && a.pb.cost_tv) {
        if (a.mode == 0) ROLLIN_LAUNCH(1, 0, true); else ROLLIN_LAUNCH(1, 1, true);
    } else if (a.pb.model == 1) {
        if (a.mode == 0) ROLLIN_LAUNCH(1, 0, false); else ROLLIN_LAUNCH(1, 1, false);
    } else {
        if (a.mode == 0) ROLLIN_LAUNCH(2, 0, false); else ROLLIN_LAUNCH(2, 1, false);
    }
#undef ROLLIN_LAUNCH
}

// =====================================================================================================
// linearize_kernel: one wavefront per (trajectory, time step); writes the tile record (layout.h).
// =====================================================================================================
__global__ __launch_bounds__(256) void linearize_kernel(LinArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15;
    const StateDev &st = a.st;
    const ProblemDev &pb = a.pb;
    const int N = st.N;
    const int nchunk = (N + 1 + 3) / 4;
    const int c = blockIdx.x / nchunk;
    const int t = (blockIdx.x - c * nchunk) * 4 + w;
    if (t > N) return;
    int b, slot;
    if (a.mode == 0) {
        b = c;
        if (st.status[b] != ST_RUNNING) return;
        slot = b * (st.E + 1) + st.slot_nom[b];
    } else {
        b = c / st.E;
        const int k = c - b * st.E;
        if (!st.ls_active[b]) return;
        slot = cand_slot(b, k, st.slot_nom[b], st.E);
    }
    const double *__restrict__ xp = st.xs + (long)slot * st.x_stride + (long)t * XSTR;
    const double *__restrict__ up = st.us + (long)slot * st.u_stride + (long)t * USTR;
    double *__restrict__ tp = st.tiles + tile_slot(st, b, slot) * st.tile_stride + (long)t * TSTRIDE;
    int dom = 0;
    if (t == N) {                                                    // terminal: h, h_x, h_xx   (ileqg.jl:314-316)
        if (pb.model == 1) {
            for (int e = l; e < 144; e += 64) tp[TT_Q + e] = pb.Qf[e];
            double part = 0.0;
            if (l < 12) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 12; ++q) acc += pb.Qf[l * 12 + q] * xp[q];
                const double qv = acc + pb.qvf[l];
                tp[TT_QV + l] = qv;
                part = xp[l] * (0.5 * acc + pb.qvf[l]);
            }
            const double tot = wave_sum(part);
            if (l == 0) tp[TT_q] = tot + pb.q0f;
        } else {
            for (int e = l; e < 144; e += 64) tp[TT_Q + e] = 0.0;
            if (l < 12) tp[TT_QV + l] = 0.0;
            if (l == 0) tp[TT_q] = pb.pl_h;
        }
        return;
    }
    const int kc = pb.cost_tv ? t : 0;
    if (pb.model == 1) {
        // f_x = A + diag(3 kappa x^2), f_u = B    (:303, :308)
        const double *__restrict__ C = pb.Ctab + (long)kc * 256;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = 64 * r + l, i = e >> 4;
            double val = pb.Zt[e];
            if (pb.kappa != 0.0 && (e & 15) == i) val += 3.0 * pb.kappa * (xp[i] * xp[i]);
            tp[TS_REG(r, l)] = val;
            tp[TS_REG(3 + r, l)] = (j < 12) ? C[e] : 0.0;            // c_xx   (:298); columns 12..15 are dead slots
        }
        tp[TS_REG(6, l)] = C[192 + l];                               // [c_ux | c_uu]   (:300-301)
        double part = 0.0;
        if (l < 16) {                                                // [c_x | c_u] = C [x;u] + [qv;rv]   (:297, :299)
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 12; ++q) acc += C[l * 16 + q] * xp[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += C[l * 16 + 12 + q] * up[q];
            const double lin = pb.lin[(long)kc * 16 + l];
            tp[TS_QR + l] = acc + lin;
            const double xu = (l < 12) ? xp[l] : up[l - 12];
            part = xu * (0.5 * acc + lin);
        }
        const double tot = wave_sum(part);
        if (l == 0) { tp[TS_q] = tot + pb.q0[kc]; tp[TS_PAD] = 0.0; }   // c   (:296)
    } else {
        // power-law family: every derivative is diagonal
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = 64 * r + l, i = e >> 4, cc = e & 15;
            double val = 0.0, cv = 0.0;
            if (i < pb.n) {
                if (cc == i) {
                    val = pb.pl_a * powchk(xp[i], pb.pl_a - 1.0, dom);
                    cv = pb.pl_cx * pb.pl_p * (pb.pl_p - 1.0) * powchk(xp[i], pb.pl_p - 2.0, dom);
                } else if (cc == 12 + i) val = pb.pl_b * powchk(up[i], pb.pl_b - 1.0, dom);
            }
            tp[TS_REG(r, l)] = val;
            tp[TS_REG(3 + r, l)] = cv;
        }
        {
            const int gg = l >> 4;
            double val = 0.0;
            if (j == 12 + gg) val = (gg < pb.m) ? pb.pl_cu * pb.pl_pu * (pb.pl_pu - 1.0) * powchk(up[gg], pb.pl_pu - 2.0, dom) : 1.0;
            tp[TS_REG(6, l)] = val;
        }
        double part = 0.0;
        if (l < 16) {
            double val = 0.0;
            if (l < pb.n) {
                val = pb.pl_cx * pb.pl_p * powchk(xp[l], pb.pl_p - 1.0, dom);
                part = pb.pl_cx * powchk(xp[l], pb.pl_p, dom);
            } else if (l >= 12 && l - 12 < pb.m) {
                val = pb.pl_cu * pb.pl_pu * powchk(up[l - 12], pb.pl_pu - 1.0, dom);
                part = pb.pl_cu * powchk(up[l - 12], pb.pl_pu, dom);
            }
            tp[TS_QR + l] = val;
        }
        const double tot = wave_sum(part);
        if (l == 0) { tp[TS_q] = tot; tp[TS_PAD] = 0.0; }
    }
    if (__ballot(dom != 0) != 0ull && l == 0) {
        if (a.mode == 0) { st.status[b] = 4; st.value[b] = INFINITY; }
        else st.flag_c[c] = 2;
    }
}

void launch_linearize(const LinArgs &a, hipStream_t s) {
    const int ntraj = (a.mode == 0) ? a.st.B : a.st.B * a.st.E;
    if (ntraj <= 0) return;
    const int nchunk = (a.st.N + 1 + 3) / 4;
    hipLaunchKernelGGL(linearize_kernel, dim3(ntraj * nchunk), dim3(256), 0, s, a);
}

#endif  // PART_ROLL
// =====================================================================================================
// per-sample control flow (one thread per sample)
// =====================================================================================================
__device__ __forceinline__ void init_state_body(const StateDev &st, const OptsDev &op, const double *theta_in, const int b);
#if RAT_PART & PART_ROLL
__global__ void init_state_kernel(StateDev st, OptsDev op, const double *theta_in) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < 2 * CTR_RING) st.counters[b] = 0;
    if (b >= st.B) return;
    init_state_body(st, op, theta_in, b);
}
#endif  // PART_ROLL
__device__ __forceinline__ void init_state_body(const StateDev &st, const OptsDev &op, const double *theta_in, const int b) {
    st.theta[b] = theta_in[b];
    st.mu[b] = 0.0;                       // initialize! sets mu = 0.0, Delta = Delta_0   (ileqg.jl:216)
    st.delta[b] = op.delta_0;
    st.value[b] = INFINITY;
    st.d_cur[b] = INFINITY;               // :217
    st.eps_init[b] = op.eps_init;         // :219
    st.ls_eps[b] = op.eps_init;
    st.status[b] = ST_RUNNING;
    st.iter[b] = 0;                       // :218
    st.ls_active[b] = 0;
    st.ls_count[b] = 0;
    st.slot_nom[b] = 0;
    st.n_ls[b] = 0;
    st.hist_n[b] = 0;
    st.lsel[b] = 0;
    st.spec_st[b] = 0;
    st.acc0[b] = 0;
}

// Start of the next step! (ileqg.jl:598-613) for a sample whose gain sweep has already been run speculatively:
// iter += 1, adopt its gains / mu / Delta (or its failure), and enter line_search! at eps_init (:502).
// The words a commit reads, fetched by its caller together with whatever else it needs (one lane runs this: every dependent load is an L2
// round trip of ~0.7 us on the sample's critical path)
struct CommitWords { int sp, iter, lsel; double mu_spec, delta_spec, eps_init; };
__device__ __forceinline__ CommitWords commit_words(const StateDev &st, int b) {
    CommitWords w;
    w.sp = xld(&st.spec_st[b]); w.iter = xld(&st.iter[b]); w.lsel = xld(&st.lsel[b]);
    w.mu_spec = xld(&st.mu_spec[b]); w.delta_spec = xld(&st.delta_spec[b]); w.eps_init = xld(&st.eps_init[b]);
    return w;
}
__device__ __forceinline__ bool commit_spec(const StateDev &st, int b, const CommitWords &w) {
    st.spec_st[b] = 0;
    st.iter[b] = w.iter + 1;                               // :599
    st.mu[b] = w.mu_spec;
    st.delta[b] = w.delta_spec;
    if (w.sp != 1) {                                       // @assert isposdef(M) (:366) / mu-restart divergence
        st.status[b] = (w.sp == 2) ? 2 : 5;
        st.value[b] = INFINITY;
        st.ls_active[b] = 0;
        return false;
    }
    st.lsel[b] = w.lsel ^ 1;                               // ileqg.L_array <- L of the gain sweep (:380)
    st.ls_eps[b] = w.eps_init;
    st.ls_count[b] = 0;
    st.ls_active[b] = 1;
    return true;
}

// after initialize!: samples that survived the open-loop sweep start step! number 1 with the speculative gains
__device__ __forceinline__ void commit_init_body(const StateDev &st, const int b) {
    const int stat = xld(&st.status[b]);
    const CommitWords w = commit_words(st, b);             // (in flight with the status word)
    if (stat != ST_RUNNING) { st.spec_st[b] = 0; return; }
    if (w.sp != 0) commit_spec(st, b, w);
}
#if RAT_PART & PART_ROLL
__global__ void commit_init_kernel(StateDev st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= st.B) return;
    commit_init_body(st, b);
}
void launch_commit_init(const StateDev &st, hipStream_t s) {
    hipLaunchKernelGGL(commit_init_kernel, dim3((st.B + 255) / 256), dim3(256), 0, s, st);
}
#endif  // PART_ROLL

__device__ __forceinline__ int uniform_load(const int *p) { return __builtin_amdgcn_readfirstlane(__atomic_load_n(p, __ATOMIC_RELAXED)); }
__device__ __forceinline__ double uniform_load_f64(const double *p) { return readlane_f64(*(const volatile double *)p, 0); }
// Does the sequential rule of line_search! reach its decision (accept, forced accept, failure) within the first kmax candidates of this
// round?  The loop of ls_select_body without its side effects; uniform over the wavefront (every lane reads the same words).
__device__ __forceinline__ bool ls_settled_within(const StateDev &st, const OptsDev &op, const int b, const int kmax) {
    if (!uniform_load(&st.ls_active[b])) return true;
    const double cur = uniform_load_f64(&st.value[b]);
    double eps = uniform_load_f64(&st.ls_eps[b]);
    const int c0 = b * st.E;
    for (int k = 0; k < kmax; ++k) {
        const int fl = uniform_load(&st.flag_c[c0 + k]);
        if (fl == 2) return true;
        if (fl == 1) { eps *= op.lambda; continue; }
        const double nv = uniform_load_f64(&st.value_c[c0 + k]);
        if (isapprox_default(nv, cur) || nv < cur) return true;
        eps *= op.lambda;
        if (eps < op.eps_min) return true;
    }
    return false;
}

// Replays the sequential rule of line_search! (ileqg.jl:504-581) over the E speculatively evaluated
// candidates of each sample, then the convergence test of solve! (:642-653).
// Counters of round `slot`: [2*slot] samples still inside their line search, [2*slot+1] samples still running.
// ctr: the round's counter pair, or null (fused solve: nobody polls)
__device__ __forceinline__ void ls_select_body(const StateDev &st, const OptsDev &op, const int b, int *ctr) {
    // One lane runs this: every per-sample scalar it can need is fetched up front, so that the loads are in flight together (a
    // chain of a dozen dependent L2 round trips otherwise: ~10k cycles per line-search decision inside the fused solve).
    const int active = xld(&st.ls_active[b]);
    const int c0 = b * st.E;
    const double eps_in = xld(&st.ls_eps[b]), cur = xld(&st.value[b]), mu_b = xld(&st.mu[b]);
    const int count_in = xld(&st.ls_count[b]), nls_in = xld(&st.n_ls[b]), nom = xld(&st.slot_nom[b]);
    CommitWords cw = commit_words(st, b);                      // (what commit_spec would otherwise fetch behind the decision: a second round trip)
    const int spec = cw.sp, iter_b = cw.iter;
    const int hn_in = st.hist ? st.hist_n[b] : 0;
    const int fl0 = xld(&st.flag_c[c0]);
    const double nv0 = xld(&st.value_c[c0]), dc0 = xld(&st.d_c[c0]);
    st.acc0[b] = 0;                                            // (the next round's candidates start unpruned)
    if (!active) return;
    double eps = eps_in;
    int count = count_in, hn = hn_in;
    int chosen = -1;
    bool failed = false;
    for (int k = 0; k < st.E; ++k) {
        const int c = c0 + k;
        count++;                                               // :505
        const int fl = (k == 0) ? fl0 : st.flag_c[c];
        if (fl == 2) { failed = true; break; }                 // exception outside the try (App. B.8)
        if (fl == 1) { eps *= op.lambda; continue; }           // :529-535
        const double nv = (k == 0) ? nv0 : st.value_c[c];
        if (st.hist) {
            if (hn < st.hist_cap) {
                st.hist[((long)b * st.hist_cap + hn) * 2] = eps;
                st.hist[((long)b * st.hist_cap + hn) * 2 + 1] = nv - cur;          // :537
            }
            st.hist_n[b] = ++hn;
        }
        if (isapprox_default(nv, cur) || nv < cur) { chosen = k; break; }           // :538
        eps *= op.lambda;                                      // :557
        if (eps < op.eps_min) { chosen = k; break; }           // :558 forced accept of the candidate just evaluated
    }
    st.n_ls[b] = nls_in + ((chosen >= 0 || failed) ? (count - count_in) : st.E);
    if (failed) {
        st.status[b] = 4; st.value[b] = INFINITY; st.ls_active[b] = 0; st.spec_st[b] = 0;
        return;
    }
    if (chosen < 0) {
        st.spec_st[b] = 0;                                     // candidate 0 was not accepted: its gain sweep is void
        if (count > 4000) { st.status[b] = 7; st.value[b] = INFINITY; st.ls_active[b] = 0; return; }
        st.ls_eps[b] = eps;
        st.ls_count[b] = count;
        if (ctr) { atomicAdd(&ctr[0], 1); atomicAdd(&ctr[1], 1); }
        return;
    }
    // accept (:539-555 / :559-575)
    const int c = c0 + chosen;
    const double d_new = (chosen == 0) ? dc0 : st.d_c[c];
    st.d_cur[b] = d_new;
    st.value[b] = (chosen == 0) ? nv0 : st.value_c[c];
    st.slot_nom[b] = (chosen < nom) ? chosen : chosen + 1;     // x_array, l_array (and their tiles) <- candidate
    st.ls_active[b] = 0;
    if (op.adaptive) {                                         // :582-591
        if (count == 1) cw.eps_init = fmin(op.eps_init, eps / op.lambda);
        else {
            while (eps < op.eps_min) eps = eps / op.lambda;
            cw.eps_init = eps;
        }
        st.eps_init[b] = cw.eps_init;
    }
    if (op.d > d_new && mu_b <= op.mu_min) { st.status[b] = 0; st.spec_st[b] = 0; }                         // converged  (:642)
    else if (iter_b == op.iter_max) { st.status[b] = 3; st.spec_st[b] = 0; }                                // iter_max   (:648)
    else if (chosen == 0 && spec != 0) {
        if (commit_spec(st, b, cw) && ctr) atomicAdd(&ctr[1], 1);  // next step! already has its gain sweep: straight to line search
    } else {
        st.spec_st[b] = 0;
        if (ctr) atomicAdd(&ctr[1], 1);                        // next round runs the plain gain sweep for this sample
    }
}

#if RAT_PART & PART_ROLL
__global__ void ls_select_kernel(StateDev st, OptsDev op, int slot) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) {                          // clear the ring entry two rounds ahead (stream order makes this race-free)
        const int z = (slot + 2) % CTR_RING;
        st.counters[2 * z] = 0;
        st.counters[2 * z + 1] = 0;
    }
    if (b >= st.B) return;
    ls_select_body(st, op, b, st.counters + 2 * slot);
}

#endif  // PART_ROLL
// =====================================================================================================
// solve_fused_kernel: the COMPLETE solve! of one theta-sample in one persistent wavefront (E = 1).
//   initialize! (open-loop rollout + linearise, open-loop policy evaluation)               ileqg.jl:214-236
//   while running: step! = gain sweep (mu restarts inside), then line_search!: closed-loop rollout + linearise,
//                  policy evaluation, accept rule; convergence / iter_max tests           ileqg.jl:598-613, 494-592, 640-654
// The phases are the SAME device functions the per-phase kernels wrap (identical arithmetic, identical results); what
// the fusion removes is everything between them: ten launches per 2-iteration solve, each with its dispatch latency, a
// prologue of dependent state loads (~8k cycles measured), a tail in which the slowest wave holds the grid, and the host
// round polling.  Samples run unsynchronised, so their tile-store bursts no longer hit HBM in lockstep.  Per-sample
// control state stays in the StateDev arrays (each phase boundary is a handful of L2 round trips); a wave's own stores
// are visible to its later loads through the CU's write-through L1 after a workgroup-scope fence (vmcnt(0)).
// =====================================================================================================
#define PHASE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")

#ifdef RAT_DIAG_PHASES
#define PHASE_MARK() do { if (threadIdx.x == 0 && blockIdx.x < 8 && fa.sw.dump && dg_pi < 40) \
        fa.sw.dump[256 + blockIdx.x * 40 + dg_pi] = (double)(__builtin_readcyclecounter() - dg_t0); ++dg_pi; } while (0)
#else
#define PHASE_MARK() do {} while (0)
#endif

// initialize!'s open-loop trajectory, rolled out once per (x_0, u_array) by the driver (FusedArgs.init_*): one wave copies the parts of its
// slot the tile-free kernels read -- the state and control histories, the [c_x | c_u | c] row of every step record (positions TS_QR ..
// TSTRIDE - 1) and the terminal tile -- into the sample's nominal slot: the bits rollin_body<.., 0, .., NOTILE> would have stored there.
__device__ __forceinline__ void copy_initial(const StateDev &st, const double *__restrict__ init_x, const double *__restrict__ init_u,
                                             const double *__restrict__ init_t, const int b) {
    const int l = threadIdx.x & 63;
    const int N = st.N;                                               // <= ROLLIN_NST (the tile-free geometries are the staged ones)
    const int slot = b * (st.E + 1);                                 // (init_state_body: slot_nom = 0)
    double *__restrict__ xo = st.xs + (long)slot * st.x_stride;
    double *__restrict__ uo = st.us + (long)slot * st.u_stride;
    double *__restrict__ tile0 = st.tiles + tile_slot(st, b, slot) * st.tile_stride;
    constexpr int RW = TSTRIDE - TS_QR;                               // the [c_x | c_u], c, pad row of a step record
    constexpr int QX = ((ROLLIN_NST + 1) * XSTR + 63) / 64, QU = (ROLLIN_NST * USTR + 63) / 64, QR = (ROLLIN_NST * RW + 63) / 64, QT = (TTERM + 63) / 64;
    const int nx = (N + 1) * XSTR, nu = N * USTR, nr = N * RW;
    // every load first (clamped addresses, ~35 in flight per lane), then the stores: one round trip instead of one per element
    double rx[QX], ru[QU], rr[QR], rt[QT];
    long orr[QR];
#pragma unroll
    for (int q = 0; q < QX; ++q) { const int e = l + 64 * q; rx[q] = init_x[e < nx ? e : 0]; }
#pragma unroll
    for (int q = 0; q < QU; ++q) { const int e = l + 64 * q; ru[q] = init_u[e < nu ? e : 0]; }
#pragma unroll
    for (int q = 0; q < QR; ++q) {
        const int e = l + 64 * q, ec = e < nr ? e : 0, t = ec / RW;
        orr[q] = (long)t * TSTRIDE + TS_QR + (ec - t * RW);
        rr[q] = init_t[orr[q]];
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) { const int e = l + 64 * q; rt[q] = init_t[(long)N * TSTRIDE + (e < TTERM ? e : 0)]; }
#pragma unroll
    for (int q = 0; q < QX; ++q) { const int e = l + 64 * q; if (e < nx) xo[e] = rx[q]; }
#pragma unroll
    for (int q = 0; q < QU; ++q) { const int e = l + 64 * q; if (e < nu) uo[e] = ru[q]; }
#pragma unroll
    for (int q = 0; q < QR; ++q) { const int e = l + 64 * q; if (e < nr) tile0[orr[q]] = rr[q]; }
#pragma unroll
    for (int q = 0; q < QT; ++q) { const int e = l + 64 * q; if (e < TTERM) tile0[(long)N * TSTRIDE + e] = rt[q]; }
}

// DUALF: where the next step!'s gain sweep reads the very tiles a policy evaluation is about to read -- initialize!'s sweep and the
// first step!, a line-search candidate and the step! that follows its acceptance -- ONE pass runs both recursions in the wave
// (sweep_dual_body: two independent dependency chains interleaved in one basic block, tiles read once; every expression is the one
// of the separate sweeps, so results are bit-identical).  The gains of the second recursion are committed by the accept rule
// (commit_spec) or dropped.  A candidate whose acceptance would END the solve (d < d_tol with mu at its floor, or iter_max) gets the
// plain policy evaluation: nothing would consume the gains.
__device__ __forceinline__ void gather_body(const StateDev &st, const int b, double *value, int *status, int *iters, int *ls_evals,
                                            double *cost, double kl_bound);

// OCC2: the compiler is held to 256 registers so that TWO samples share a SIMD (only offered without DUALF / STG: one recursion per pass,
// 2 KB of LDS per wave) -- the direct test of "hide a wave's dependency stalls with a second sample" for batches beyond one per SIMD.
// MAT: the tile-free geometry with its tile records put back (rollouts write them, sweeps load them): the materialised formulation SURVEY 8d
// words its byte model on, kept as a measured variant (bit-identical: fx_diag gives a record the bits the fly sweeps form in registers).
template <int MODEL, bool CTV, int WM, bool DUALF, bool STG, bool OCC2 = false, bool MAT = false>
__global__ __launch_bounds__(64, OCC2 ? 2 : 1) void solve_fused_kernel(FusedArgs fa) {
    const int b = blockIdx.x;
    const StateDev &st = fa.sw.st;
#ifdef RAT_DIAG_PHASES
    const unsigned long long dg_t0 = __builtin_readcyclecounter();
    int dg_pi = 0;
#endif
    __shared__ double wls[DUALF ? WLS_DUAL : WLS_SWEEP];
    __shared__ double shxu[16];
    __shared__ double stg[STG ? STG_DOUBLES : 1];
    // The headline geometry (LQ family, paired recursions, staged rollouts) keeps NO tiles in HBM: every sweep of the solve forms f_x | f_u
    // and the cost Hessian of step t from x_t and the problem tables (FLY sweeps, as on the speculative path), so the rollouts store per
    // step only [x_t; u_t] and the [c_x | c_u | c] row -- 0.3 KB instead of 3.4 KB; the registers of a sweep hold the bits a record
    // would have delivered (fx_diag), so results are identical to the tile-materialising paths.
    // (OCC2: tile-free as well -- its rollouts fetch their operands from HBM step by step instead of staging them, the second wave of the
    //  SIMD covers the latency; 2 KB of LDS per wave)
    constexpr int FLYF = (MODEL == 1 && ((DUALF && STG && !OCC2) || (OCC2 && !DUALF && !STG)) && !MAT) ? (CTV ? 2 : 1) : 0;
    constexpr bool NT = FLYF != 0;
    // the sample's own wave initialises its state and, at the end, writes its outputs: a batch is ONE launch
    if (threadIdx.x == 0) init_state_body(st, fa.sw.op, fa.theta_in, b);
    PHASE_FENCE();
    {
        if (NT && fa.init_x) {
            copy_initial(st, fa.init_x, fa.init_u, fa.init_t, b);
        } else {
            RolloutArgs ra = fa.ro; ra.mode = 0;
            rollin_body<MODEL, 0, CTV, STG, true, OCC2 ? OCC2_PREFETCH : ROLLIN_PREFETCH, NT>(ra, b, shxu, stg);
        }
        PHASE_MARK();
        PHASE_FENCE();
        PHASE_MARK();
        if (DUALF) {
            SweepArgs sa = fa.sw; sa.mode = 6;
            sweep_dual_body<WM, false, FLYF>(sa, b, wls);
            PHASE_MARK();
            PHASE_FENCE();
            if (threadIdx.x == 0) commit_init_body(st, b);
            PHASE_FENCE();
            PHASE_MARK();
        } else {
            SweepArgs sa = fa.sw; sa.mode = 2;
            sweep_body<false, false, WM, false, OCC2 ? OCC2_SWZ : 0, FLYF>(sa, b, wls);
            PHASE_MARK();
            PHASE_FENCE();
            PHASE_MARK();
        }
    }
    for (int guard = 0; guard < fa.max_rounds; ++guard) {
        // (both control words are requested before either is waited for: one L2 round trip per loop turn instead of two)
        const int v_stat = __atomic_load_n(&st.status[b], __ATOMIC_RELAXED), v_act = __atomic_load_n(&st.ls_active[b], __ATOMIC_RELAXED);
        if (__builtin_amdgcn_readfirstlane(v_stat) != ST_RUNNING) break;
        if (!__builtin_amdgcn_readfirstlane(v_act)) {        // step!: solve_approximate_dp!  (ileqg.jl:598-613)
            SweepArgs sa = fa.sw; sa.mode = 0;
            sweep_body<true, false, WM, false, OCC2 ? OCC2_SWZ : 0, FLYF>(sa, b, wls);
            PHASE_MARK();
            PHASE_FENCE();
            PHASE_MARK();
            continue;
        }
        {                                                    // one candidate of line_search!  (ileqg.jl:504-581)
            RolloutArgs ra = fa.ro; ra.mode = 1;
            rollin_body<MODEL, 1, CTV, STG, true, OCC2 ? OCC2_PREFETCH : ROLLIN_PREFETCH, NT>(ra, b, shxu, stg);
            PHASE_MARK();
            PHASE_FENCE();
            PHASE_MARK();
            bool pair = DUALF;
            if (DUALF) {                                     // would accepting this candidate end solve!?  (ileqg.jl:642-653)
                const double v_dc = *(const volatile double *)&st.d_c[b], v_mu = *(const volatile double *)&st.mu[b];
                const int v_it = __atomic_load_n(&st.iter[b], __ATOMIC_RELAXED);            // (three loads in flight together)
                const double dc = readlane_f64(v_dc, 0), mu = readlane_f64(v_mu, 0);
                const bool ends = (fa.sw.op.d > dc && mu <= fa.sw.op.mu_min) || __builtin_amdgcn_readfirstlane(v_it) == fa.sw.op.iter_max;
                pair = !ends;
            }
            if (pair) {
                SweepArgs sa = fa.sw; sa.mode = 7;
                sweep_dual_body<WM, true, FLYF>(sa, b, wls);
            } else {
                SweepArgs sa = fa.sw; sa.mode = 1;
                sweep_body<false, false, WM, true, OCC2 ? OCC2_SWZ : 0, FLYF>(sa, b, wls);
            }
            PHASE_MARK();
            PHASE_FENCE();
            if (threadIdx.x == 0) ls_select_body(st, fa.sw.op, b, nullptr);
            PHASE_FENCE();
            PHASE_MARK();
        }
    }
    PHASE_FENCE();
    if (threadIdx.x == 0) gather_body(st, b, fa.out_value, fa.out_status, fa.out_iters, fa.out_ls, fa.out_cost, fa.kl_bound);
}


#if RAT_PART & PART_FUSED
void launch_solve_fused(const FusedArgs &fa, hipStream_t s) {
    const int B = fa.sw.st.B;
    if (B <= 0) return;
    const dim3 grid(B), block(64);
    const int wm = fa.sw.pb.W_tv ? 1 : (fa.sw.pb.W_diag ? 2 : 0);
    const bool stg = fa.sw.pb.model == 1 && fa.sw.st.N <= ROLLIN_NST;
#define FUSED_LAUNCH(M, C, W) do { \
        if (fa.occ2) hipLaunchKernelGGL((solve_fused_kernel<M, C, W, false, false, true>), grid, block, 0, s, fa); \
        else if (fa.mat && fa.dual && stg && M == 1 && !C) hipLaunchKernelGGL((solve_fused_kernel<1, false, W, true, true, false, true>), grid, block, 0, s, fa); \
        else if (fa.dual && stg && M == 1) hipLaunchKernelGGL((solve_fused_kernel<M, C, W, true, M == 1>), grid, block, 0, s, fa); \
        else hipLaunchKernelGGL((solve_fused_kernel<M, C, W, true, false>), grid, block, 0, s, fa); } while (0)
    if (fa.sw.pb.model == 1) {
        if (fa.sw.pb.cost_tv) { if (wm == 1) FUSED_LAUNCH(1, true, 1); else if (wm == 2) FUSED_LAUNCH(1, true, 2); else FUSED_LAUNCH(1, true, 0); }
        else { if (wm == 1) FUSED_LAUNCH(1, false, 1); else if (wm == 2) FUSED_LAUNCH(1, false, 2); else FUSED_LAUNCH(1, false, 0); }
    } else {
        if (wm == 1) FUSED_LAUNCH(2, false, 1); else if (wm == 2) FUSED_LAUNCH(2, false, 2); else FUSED_LAUNCH(2, false, 0);
    }
#undef FUSED_LAUNCH
}

#endif  // PART_FUSED
// =====================================================================================================
// solve_block_kernel: the complete solve! of one theta-sample by a WORKGROUP of wavefronts -- one wave per speculative line-search
// candidate (E of them) plus, when GW, one wave that runs the gain sweeps:
//   wave 0        initialize!'s rollout + open-loop policy evaluation; candidate 0 of every line-search round; the accept rule
//   waves 1..E-1  candidates 1..E-1 (eps_k = eps lambda^k): rollout + linearise, policy evaluation       (ileqg.jl:504-536)
//   wave E (GW)   solve_approximate_dp! of the NEXT step!, speculatively on candidate 0's tiles while the candidates are evaluated
//                 (what step! would compute if candidate 0 is accepted, App. B.1; dropped otherwise), the first gain sweep beside
//                 initialize!'s sweep, and the plain gain sweep whenever no speculative one is valid
// The phases are the SAME device functions as everywhere else (rollin_body, sweep_body, ls_select_body): results are bit-identical
// to solve_fused_kernel and to the round-based path.  Where solve_fused_kernel pairs a policy evaluation with the following gain
// sweep as two recursions inside ONE wave (sweep_dual_body, 342 registers: one wave per SIMD), this kernel puts them on two waves
// of <= 256 registers each, so that
//   * a sample's evaluation and gain recursions run CONCURRENTLY on two SIMDs when the batch leaves SIMDs free (strong scaling:
//     1024 / N samples per GPU) -- the critical path of a 2-iteration solve drops from 5 sweeps' worth to 3;
//   * two waves per SIMD fit, so at a full batch the hardware interleaves two independent waves per SIMD;
//   * E > 1 candidates are evaluated inside the launch (line searches that really backtrack, BASELINE config 3).
// Waves meet at workgroup barriers between phases; per-sample control state stays in the StateDev arrays as in the fused kernel.
// =====================================================================================================
// Where the waves of a two-wave workgroup land (E = 1; tools/ubench/placement.hip, profiles/r02_wave_placement.md).  The dispatcher deals
// workgroups evenly over the CUs and puts a workgroup's waves on consecutive SIMDs of the cyclic order 0 -> 2 -> 1 -> 3, but the NEXT
// workgroup on the same CU starts on the SIMD the previous one's wave 1 sits on: with two two-wave workgroups per CU (512 samples) one
// SIMD hosts two waves and one hosts none.  PAD4 launches four waves per workgroup instead -- one per SIMD, always -- draws a ticket
// from a per-CU counter (hardware CU id, one relaxed atomic per workgroup) and keeps the pair of SIMDs the ticket's parity names
// ({0, 2} / {1, 3}); the other two waves exit at once.  Consecutive workgroups of a CU so use complementary SIMD pairs: 512 samples run
// in 0.339 ms instead of 0.413 ms.  (The host uses it up to two workgroups per CU: the register slots of the waves that exit stay
// charged to their workgroup until it ends.)  The ticket is a placement heuristic: whatever it returns, results are the same.
__device__ __forceinline__ int hw_cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID [3:0]
    return (int)(((xcc & 15u) << 8) | ((hw >> 8) & 255u));                // < CENSUS_SLOTS
}
__device__ __forceinline__ int hw_simd_id() { return (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u); }

template <int MODEL, bool CTV, int WM, int NW, bool GW, bool STG, bool PAD4>
__global__ __launch_bounds__(PAD4 ? 256 : 64 * NW, 2) void solve_block_kernel(FusedArgs fa) {
    static_assert(!PAD4 || NW == 2, "PAD4 is the two-wave (E = 1) geometry");
    // LAZY (NW = 8 with a gain wave: E = 8): a CU holds eight waves of this kernel's 256 registers, so the ninth wave a gain sweep beside
    // eight evaluations would need does not exist.  Wave 7 rolls candidate 7 out with the others, then runs the GAIN sweeps while waves
    // 0..6 evaluate candidates 0..6; candidate 7 (eps lambda^7) is evaluated afterwards, by wave 7, only when the accept rule did not
    // settle on one of the first seven (ls_settled_within) -- the rule reads candidates in order and stops at the first it accepts, so
    // what it reads, and every result, is unchanged.
    constexpr bool LAZY = GW && NW == 8;
    constexpr int E = LAZY ? NW : (GW ? NW - 1 : NW);          // candidates
    constexpr int WG = GW ? NW - 1 : 0;          // the wave that runs gain sweeps
    constexpr int HWAVES = PAD4 ? 4 : NW;
    // two waves, LQ family, staged operands: the rollouts are split over both waves (rollrec_body / rolllin_body)
    constexpr bool SPLIT = (NW == 2) && GW && STG && (MODEL == 1);
    // LQ family with staged rollouts (N <= ROLLIN_NST): no tiles in HBM either (see solve_fused_kernel) -- the rollouts (split or one wave
    // per candidate) store [x; u] and the cost-gradient row, every sweep forms f_x | f_u and the Hessian in registers
    constexpr int FLYB = (MODEL == 1 && STG) ? (CTV ? 2 : 1) : 0;
    constexpr bool NTB = FLYB != 0;
    const int b = blockIdx.x;
    const int hwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const StateDev &st = fa.sw.st;
    __shared__ double wls_all[HWAVES][WLS_SWEEP];
    __shared__ double shxu_all[HWAVES][16];
    __shared__ double stg[STG ? STG_DOUBLES : 1];
    __shared__ double xu[SPLIT ? XU_DOUBLES : 1];
    __shared__ double aclring[SPLIT ? ACL_DOUBLES : 1];
    __shared__ int prog, pprog, lpool, rdone;
    __shared__ unsigned long long d_acc[2];
    constexpr bool PRUNEB = (E > 1) && (FLYB != 0) && !SPLIT;      // speculation pruned (FusedArgs.sw.prune): see the line-search round below
    int rnd = 0;                                                  // line-search rounds so far (every wave counts alike)
    // PSW2 (the two-wave geometry, LQ family, staged): the evaluation that ENDS the solve has the sample's gain wave idle beside it -- the two
    // run it time-parallel (psweep.h, P = 2: 19 + 15 ordinary steps and a hop instead of 50 steps).  Equal to rounding, switch block_psw.
    constexpr bool PSW2 = SPLIT && !CTV && (WM != 1);
    __shared__ PswSharedT<PSW2 ? 2 : 1> psh2;
    __shared__ double wls_psw[PSW2 ? 2 : 1][PSW2 ? WLS_PSW : 1];
    if (PSW2 && threadIdx.x == 0) { psh2.flag[0] = 0; psh2.flag[PSW2 ? 1 : 0] = 0; psh2.bar = 0; psh2.last_rc = 0; psh2.lastP = 0; }
    constexpr bool HELP = SPLIT && PAD4;         // fa.helpers: the two waves a padded workgroup does not need stay as linearise helpers
    int epoch = 0;
    if (threadIdx.x == 0) { prog = 0; pprog = 0; rdone = 0; }
    double *const wls = wls_all[hwave], *const shxu = shxu_all[hwave];
    int wave = hwave;                            // role: 0 .. E-1 candidates, WG gain sweeps
    if (PAD4) {
        __shared__ int s_simd[4], s_role[4];
        if ((threadIdx.x & 63) == 0) s_simd[hwave] = hw_simd_id();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int ticket = atomicAdd(&fa.census[hw_cu_key()], 1);
            const int odd = ticket & 1, flip = (ticket >> 1) & 1;
            // The SIMD-derived roles need the four waves on four DISTINCT SIMDs.  The dispatcher gives that to a 256-thread workgroup
            // on an otherwise empty CU, but not when another kernel is resident on it (a second handle or stream, RCCL, a framework
            // kernel): two waves may then share a SIMD, and roles read off the SIMD ids would leave the workgroup without a leader or
            // without a gain wave.  Anything but a permutation of 0..3 falls back to roles by wave index.
            unsigned seen = 0;
            for (int w = 0; w < 4; ++w) seen |= 1u << (s_simd[w] & 3);
            const bool distinct = (seen == 0xFu);
            for (int w = 0; w < 4; ++w) {
                const int sid = s_simd[w];
                const bool in_pair = odd ? (sid == 1 || sid == 3) : (sid == 0 || sid == 2);
                const bool first = odd ? (sid == 1) : (sid == 0);
                s_role[w] = distinct ? (in_pair ? ((first != (flip != 0)) ? 0 : 1) : -1) : (w < 2 ? w : -1);
            }
            if (HELP && fa.helpers) {            // one workgroup per CU: all four SIMDs are this sample's
                int next = 2;
                for (int w = 0; w < 4; ++w) if (s_role[w] < 0) s_role[w] = next++;
            }
        }
        __syncthreads();
        wave = __builtin_amdgcn_readfirstlane(s_role[hwave]);
        // (an ended wave no longer takes part in the workgroup's barriers: s_barrier on gfx9 / CDNA counts the waves of the workgroup that
        //  are still alive -- validated on gfx950 (tools/stress_block.py); HIP leaves a barrier not reached by every thread undefined,
        //  so this early exit must be revisited for any other target)
        if (wave < 0) return;
    }
    const bool leader = (wave == 0) && ((threadIdx.x & 63) == 0);
#ifdef RAT_DIAG_PHASES
    const unsigned long long dg_t0 = __builtin_readcyclecounter();
    int dg_pi = 0;
#define BLK_MARK() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8 && fa.sw.dump && dg_pi < 40 && (wave == 0 || wave == (GW ? WG : 1))) \
        fa.sw.dump[1024 + blockIdx.x * 80 + (wave ? 40 : 0) + dg_pi] = (double)(__builtin_readcyclecounter() - dg_t0); ++dg_pi; } while (0)
#else
#define BLK_MARK() do {} while (0)
#endif
    if (leader) init_state_body(st, fa.sw.op, fa.theta_in, b);
    BLK_MARK();
    __syncthreads();
    BLK_MARK();
    const bool helpers = HELP && fa.helpers;
    const int nlin = helpers ? 3 : 1;            // linearise waves: the gain wave (+ the two spare waves)
    if (NTB && fa.init_x) {                      // initialize!'s rollout was run once for the whole batch (FusedArgs.init_*)
        if (wave == 0) copy_initial(st, fa.init_x, fa.init_u, fa.init_t, b);
    } else if (SPLIT) {                          // initialize!: open-loop rollout (wave 0) + linearise (the other waves)   (ileqg.jl:214-233)
        RolloutArgs ra = fa.ro; ra.mode = 0;
        if (helpers) {
            if (wave == 0) rollrec_body<0, HELP>(ra, b, stg, xu, &prog, epoch, d_acc);
            else rolllin_body<0, CTV, HELP, NTB>(ra, b, shxu, xu, &prog, epoch, wave - 1, nlin, d_acc);
        } else {
            if (wave == 0) rollrec_body<0, false>(ra, b, stg, xu, &prog, epoch, d_acc);
            else rolllin_body<0, CTV, false, NTB>(ra, b, shxu, xu, &prog, epoch, 0, 1, d_acc);
        }
        epoch += st.N + 2;
    } else if (wave == 0) {
        RolloutArgs ra = fa.ro; ra.mode = 0;
        rollin_body<MODEL, 0, CTV, STG, true, ROLLIN_PREFETCH, NTB>(ra, b, shxu, stg);
    }
    BLK_MARK();
    __syncthreads();
    BLK_MARK();
    if (wave == 0) {                             // open-loop policy evaluation (:234) ...
        SweepArgs sa = fa.sw; sa.mode = 2;
        sweep_body<false, false, WM, false, 0, FLYB>(sa, b, wls);
    } else if (GW && wave == WG) {               // ... beside the first step!'s gain sweep on the same tiles (speculative until initialize! succeeds)
        SweepArgs sa = fa.sw; sa.mode = 5;
        sweep_body<true, false, WM, false, 0, FLYB>(sa, b, wls);
    }
    BLK_MARK();
    __syncthreads();
    BLK_MARK();
    if (GW) {
        if (leader) commit_init_body(st, b);
        BLK_MARK();
    __syncthreads();
    BLK_MARK();
    }
    for (int guard = 0; guard < fa.max_rounds; ++guard) {
        const int v_stat = __atomic_load_n(&st.status[b], __ATOMIC_RELAXED), v_act = __atomic_load_n(&st.ls_active[b], __ATOMIC_RELAXED);
        if (__builtin_amdgcn_readfirstlane(v_stat) != ST_RUNNING) break;
        if (!__builtin_amdgcn_readfirstlane(v_act)) {        // step!: solve_approximate_dp! with no valid speculative sweep  (ileqg.jl:598-613)
            if (wave == WG) {
                SweepArgs sa = fa.sw; sa.mode = 0;
                sweep_body<true, false, WM, false, 0, FLYB>(sa, b, wls);
            }
            BLK_MARK();
    __syncthreads();
    BLK_MARK();
            continue;
        }
        if (SPLIT && fa.acl) {                                // the candidate of this line-search round in deviation form (rollacl_body)
            RolloutArgs ra = fa.ro; ra.mode = 1;
            int nom_, lsel_;
            double eps_;
            const bool act = rollout_active<1>(st, b, nom_, lsel_, eps_);       // (every wave reads the same words)
            if (act) {
                if (leader) { d_acc[0] = 0ull; d_acc[1] = 0ull; lpool = 0; }
                if (helpers) stage_shared<4>(ra, b, nom_, lsel_, stg, wave); else stage_shared<2>(ra, b, nom_, lsel_, stg, wave);
            }
            BLK_MARK();
            __syncthreads();
            BLK_MARK();
            if (act) {
                // wave 0 runs the recursion, wave 1 produces its operands ahead of it; the steps are linearised by whoever is free (a
                // counter in LDS hands out the next one): the spare waves from the start, the producer and the recursion wave once they
                // are done.  The recursion wave, which holds x_N, writes the terminal tile.
                ACL_MARK(fa.sw.dump, wave, 0, __builtin_readcyclecounter());
                if (wave == 0) rollacl_body(ra, eps_, stg, aclring, xu, &pprog, &prog, epoch);
                else if (wave == 1) rollprod_body(ra, stg, aclring, &pprog, &prog, epoch);
                ACL_MARK(fa.sw.dump, wave, 1, __builtin_readcyclecounter());
                rolllin_body<1, CTV, HELP, NTB, true>(ra, b, shxu, xu, &prog, epoch, 0, 1, d_acc, stg, eps_, &lpool, wave == 0 ? 1 : 0);
                ACL_MARK(fa.sw.dump, wave, 2, __builtin_readcyclecounter());
            }
            epoch += st.N + 2;
        } else if (SPLIT) {                                   // the candidate of this line-search round  (ileqg.jl:504-521), split over both waves
            RolloutArgs ra = fa.ro; ra.mode = 1;
            if (helpers) {
                if (wave == 0) rollrec_body<1, HELP>(ra, b, stg, xu, &prog, epoch, d_acc);
                else rolllin_body<1, CTV, HELP, NTB>(ra, b, shxu, xu, &prog, epoch, wave - 1, nlin, d_acc);
            } else {
                if (wave == 0) rollrec_body<1, false>(ra, b, stg, xu, &prog, epoch, d_acc);
                else rolllin_body<1, CTV, false, NTB>(ra, b, shxu, xu, &prog, epoch, 0, 1, d_acc);
            }
            epoch += st.N + 2;
        } else if (wave < E) {                                // candidates of this line-search round  (ileqg.jl:504-521)
            // E > 1, speculation pruned (FusedArgs.sw.prune): candidate 0 runs ahead of the wave it shares a SIMD with -- no workgroup barrier
            // between the rollouts and the evaluations: every candidate's evaluation follows its own rollout, the gain wave waits for candidate
            // 0's alone (a word in LDS) -- and when its evaluation shows that the line search settles on it, the other candidates' evaluations
            // stop (their results would never be read).  The last candidate (LAZY: the gain wave's) is rolled out only if the rule gets that far.
            RolloutArgs ra = fa.ro; ra.mode = 1;
            if (PRUNEB && fa.sw.prune) {
                if (wave == 0) __builtin_amdgcn_s_setprio(2);
                if (!(LAZY && wave == WG)) rollin_body<MODEL, 1, CTV, STG, false, ROLLIN_PREFETCH, NTB>(ra, b * E + wave, shxu, stg);
                PHASE_FENCE();                                // (the wave's own evaluation reads what it has just stored)
                if (leader) __hip_atomic_store(&rdone, rnd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                rollin_body<MODEL, 1, CTV, STG, false, ROLLIN_PREFETCH, NTB>(ra, b * E + wave, shxu, stg);
            }
        }
        BLK_MARK();
        if (!(PRUNEB && fa.sw.prune))
    __syncthreads();
        ++rnd;
    BLK_MARK();
        bool psw_done = false;
        if constexpr (PSW2) if (fa.psw_last && !helpers && wave <= 1) {
            // does accepting this candidate end solve! (:642-653)?  Then no gain sweep runs beside the evaluation: both waves evaluate.
            // (d of the candidate: written by the recursion wave before the barrier, or gathered in d_acc by the linearise waves)
            double v_dc;
            if (SPLIT && fa.acl) { const unsigned long long da0 = d_acc[0], da1 = d_acc[1]; v_dc = da1 ? NAN : sqrt(__longlong_as_double((long long)da0)); }
            else v_dc = *(const volatile double *)&st.d_c[b * E];
            const double v_mu = *(const volatile double *)&st.mu[b];
            const int v_it = __atomic_load_n(&st.iter[b], __ATOMIC_RELAXED);
            const double dc = readlane_f64(v_dc, 0), mu = readlane_f64(v_mu, 0);
            const bool ends = (fa.sw.op.d > dc && mu <= fa.sw.op.mu_min) || __builtin_amdgcn_readfirstlane(v_it) == fa.sw.op.iter_max;
            if (ends) {
                if (SPLIT && fa.acl && wave == WG && (threadIdx.x & 63) == 0) { st.d_c[b * E] = v_dc; st.flag_c[b * E] = 0; }
                SweepArgs sa = fa.sw; sa.mode = 1;
                psweep_body<false, WM, true, FLYB>(sa, b, wls_psw[wave], &psh2, fa.psw2e, wave);
                psw_done = true;
            }
        }
        if (psw_done) {
        } else if (wave < E && !(LAZY && wave == WG)) {       // their policy evaluations  (:522-536)
            SweepArgs sa = fa.sw; sa.mode = 1;
            if (PRUNEB && sa.prune && wave > 0) sweep_body<false, false, WM, true, 0, FLYB, PRUNEB>(sa, b * E + wave, wls);
            else sweep_body<false, false, WM, true, 0, FLYB>(sa, b * E + wave, wls);
            if (PRUNEB && sa.prune && wave == 0) __builtin_amdgcn_s_setprio(0);
        } else if (wave == WG) {                              // the gain wave: next step!'s sweep on candidate 0's tiles, unless accepting
            if (PRUNEB && fa.sw.prune) { spin_until(&rdone, rnd); PHASE_FENCE(); }          // candidate 0 has been rolled out
            if ((helpers || (SPLIT && fa.acl)) && (threadIdx.x & 63) == 0) {      // d of the candidate, gathered by the linearise waves: to where the accept rule reads it
                st.d_c[b * E] = d_acc[1] ? NAN : sqrt(__longlong_as_double((long long)d_acc[0]));
                st.flag_c[b * E] = 0;
            }
            if (helpers || (SPLIT && fa.acl)) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const double v_dc = *(const volatile double *)&st.d_c[b * E], v_mu = *(const volatile double *)&st.mu[b];   // it ends solve! (:642-653)
            const int v_it = __atomic_load_n(&st.iter[b], __ATOMIC_RELAXED);
            const double dc = readlane_f64(v_dc, 0), mu = readlane_f64(v_mu, 0);
            const bool ends = (fa.sw.op.d > dc && mu <= fa.sw.op.mu_min) || __builtin_amdgcn_readfirstlane(v_it) == fa.sw.op.iter_max;
            if (!ends) {
                SweepArgs sa = fa.sw; sa.mode = 4;
                sweep_body<true, false, WM, false, 0, FLYB>(sa, b, wls);
            }
        }
        BLK_MARK();
    __syncthreads();
    BLK_MARK();
        if (LAZY) {                                           // the last candidate, only if the rule gets that far (every wave reads the same words)
            if (!ls_settled_within(st, fa.sw.op, b, E - 1)) {
                if (wave == WG) {
                    if (PRUNEB && fa.sw.prune) {
                        RolloutArgs ra = fa.ro; ra.mode = 1;
                        rollin_body<MODEL, 1, CTV, STG, false, ROLLIN_PREFETCH, NTB>(ra, b * E + wave, shxu, stg);
                        PHASE_FENCE();
                    }
                    SweepArgs sa = fa.sw; sa.mode = 1;
                    sweep_body<false, false, WM, true, 0, FLYB>(sa, b * E + wave, wls);
                }
                BLK_MARK();
    __syncthreads();
                BLK_MARK();
            }
        }
        if (leader) ls_select_body(st, fa.sw.op, b, nullptr);
        BLK_MARK();
    __syncthreads();
    BLK_MARK();
    }
    BLK_MARK();
    __syncthreads();
    BLK_MARK();
    if (leader) gather_body(st, b, fa.out_value, fa.out_status, fa.out_iters, fa.out_ls, fa.out_cost, fa.kl_bound);
}

#undef BLK_MARK

// =====================================================================================================
// solve_block_psw_kernel: the whole solve! of one theta-sample by a workgroup of FOUR wavefronts on a compute unit of its own (E = 1, LQ
// family, batches of at most one sample per CU: strong-scaling shards, Nelder-Mead batches, the final solve), every Riccati sweep
// TIME-PARALLEL (psweep.h).  solve_block_kernel walks each sweep's 50 dependent steps with one wavefront while the sample's other SIMDs
// idle; here
//   [initialize!'s evaluation || first gain sweep]          two teams of two waves each (waves {0, 2} / {1, 3}), side by side
//   rollout                                                  recursion wave + three linearising waves (rollrec_body / rolllin_body, unchanged)
//   [candidate's evaluation || next step!'s gain sweep]     the two teams again
//   the evaluation that ends the solve, a plain gain sweep   one team of all four waves
// Same control flow and the same device functions for everything but the sweeps as solve_block_kernel; the sweeps agree with the
// sequential ones to rounding (boundary values handed along the chain differ by ~1e-15), so values are NOT bit-identical to the other
// paths -- identical status / iteration / line-search counts (tests/test_gpu_psweep.py), values to 1e-10.
// =====================================================================================================
// ---- two workgroups per sample ("duo"; FusedArgs.duo_stride > 0, batches of at most half a sample per CU) ----------------------------------------
// At <= n_cu / 2 samples half the compute units are dark while every sample's gain team (two waves) is the long pole of both sweep phases.
// The launch then carries TWO workgroups per sample, dealt so that partners land in one XCD (blocks i and i + 8 of a group of 16: workgroups go
// to the 8 XCDs round-robin; checked at run time from HW_REG_XCC_ID) and therefore share an L2:
//   role A (the solve)   everything solve_block_psw_kernel does except the gain sweeps; its policy evaluations as FOUR-wave teams
//   role B (gain team)   every gain sweep of the sample as a four-wave team: initialize!'s speculative one (mode 5), the next step!'s beside
//                        each candidate's evaluation (mode 4), the plain one after a rejected candidate (mode 0)
// Hand-overs go through global memory: the producer's waves drain their stores (s_waitcnt vmcnt(0): the write-through L1 has delivered them to
// the shared L2), a barrier, then ONE 64-bit word (launch epoch << 32 | sequence number; role A's posts also carry what they ask of role B) is
// stored; the consumer's leader polls it with
// agent-scope relaxed loads, and every load of partner-written data is an sc1 load (xld, device_utils.h) that misses the consumer's L1.  No
// agent-scope fence anywhere: buffer_wbl2 / buffer_inv sc1 walk the L2 (3.6 - 7 us per hand-over, tools/ubench/xwg_handoff.hip: more than
// the second compute unit saves); this way a hand-over costs 0.5 - 2 us.
// Co-residency is NOT assumed (an ordinary launch; hipLaunchCooperativeKernel costs +19 us): role B checks in with a CAS on the pair word and
// role A -- dispatched earlier: workgroups start in order -- decides once, after initialize!'s copy, whether its partner is there (same XCD):
// "duo", or "solo" = the one-workgroup schedule below, in which case a late partner leaves at once.  A resident role A never waits for a
// partner that has not checked in, a role B only waits for a role A that is running: no deadlock whatever else shares the device.
#define XC_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ long long xw_load(long long *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xw_store(long long *w, long long v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all threads: drain, barrier, the leader posts
__device__ __forceinline__ void xc_post(long long *w, const long long v) {
    XC_DRAIN();
    __syncthreads();
    if (threadIdx.x == 0) xw_store(w, v);
}
// all threads: the leader polls until *w >= v (bounded: a protocol error must not hang the device), barrier; returns the word (0 on timeout)
__device__ __forceinline__ long long xc_wait(long long *w, const long long v, long long *const bcast) {
    if (threadIdx.x == 0) {
        long long got = 0;
        for (int polls = 0; polls < (1 << 22); ++polls) {
            const long long x = xw_load(w);
            if (x - v >= 0) { got = x; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *bcast = got;
    }
    __syncthreads();
    const long long r = *bcast;
    __syncthreads();
    asm volatile("" ::: "memory");
    return r;
}

typedef int kw_v16i __attribute__((ext_vector_type(16)));
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    kw_v16i sink;
    static_assert(BYTES % 64 == 0 && BYTES <= 2048, "whole lines, immediate offsets");
#define KW_LD(off_) do { if ((off_) < BYTES) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(sink) : "s"(kp), "n"(off_) : "memory"); } while (0)
#define KW_LD4(o_) KW_LD(o_); KW_LD((o_) + 64); KW_LD((o_) + 128); KW_LD((o_) + 192)
    KW_LD4(0); KW_LD4(256); KW_LD4(512); KW_LD4(768); KW_LD4(1024); KW_LD4(1280); KW_LD4(1536); KW_LD4(1792);
#undef KW_LD4
#undef KW_LD
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" :: "s"(sink));
}

template <bool CTV, int WM>
__global__ __launch_bounds__(256, 1) void solve_block_psw_kernel(FusedArgs fa) {
    constexpr int FLYB = CTV ? 2 : 1;
    // (SEQ4: the evaluation and the gain sweep one after the other as four-wave teams instead of side by side as two-wave teams -- a
    //  diagnostic of the W(k) miscompile, see launch_solve_block_psw_tv; off)
    constexpr bool SEQ4 = false;
    // duo launches: groups of 16 blocks = 8 samples x {role A, role B}, partners 8 blocks apart (one XCD)
    const int role = (fa.duo_stride > 0) ? ((blockIdx.x >> 3) & 1) : 0;
    const int b = (fa.duo_stride > 0) ? (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) : blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const StateDev &st = fa.sw.st;
    if (b >= st.B) return;
    // The kernel arguments are 1.5 KB (24 lines) that the phases below load piece by piece into scalar registers, each piece for the first time
    // on the critical path: a cold round trip to device memory of ~2 us (4.4 us measured in front of the first rollout alone).  Wave 3, which
    // starts with nothing to do, pulls the whole block through the scalar cache once.
    if (wave == 3) kernarg_warm<sizeof(FusedArgs)>();
    __shared__ double wls_all[4][WLS_PSW];
    __shared__ double shxu_all[4][16];
    __shared__ double stg[STG_DOUBLES];
    __shared__ double xu[XU_DOUBLES];
    __shared__ double aclring[ACL_DOUBLES];
    __shared__ int prog, pprog, lpool;
    __shared__ unsigned long long d_acc[2];
    __shared__ long long xbc;                    // xc_wait's broadcast; the duo decision
    __shared__ PrlShared prl;                    // the time-parallel rollout's hand-over boxes (rollprl_body)
    __shared__ PswShared psh[3];                 // [0] the evaluation team, [1] the gain team, [2] the four-wave team (a team's barrier counter
                                                 // stays a multiple of ITS size)
    int epoch = 0;
    if (threadIdx.x == 0) { prog = 0; pprog = 0; }
    if (threadIdx.x < PRL_WAVES) prl.flag[threadIdx.x] = 0;
    if (threadIdx.x < 3 * PSW_MAXP) psh[threadIdx.x / PSW_MAXP].flag[threadIdx.x % PSW_MAXP] = 0;
    if (threadIdx.x < 3) { psh[threadIdx.x].bar = 0; psh[threadIdx.x].last_rc = 0; psh[threadIdx.x].lastP = 0; }
    double *const wls = wls_all[wave], *const shxu = shxu_all[wave];
    const bool leader = (wave == 0) && ((threadIdx.x & 63) == 0);
    const int team = wave & 1, tw = wave >> 1;          // evaluations: waves {0, 2}; gain sweeps: waves {1, 3}
    long long *const xw = fa.xw ? fa.xw + (long)b * XW_STRIDE : nullptr;
    const long long xep = (long long)fa.xepoch << 32;
    int seqA = 0, seqB = 0;
#ifdef RAT_DIAG_PHASES
    int dg_pi = 0;
    // (role A: waves 0 and 1 at every barrier; role B: wave 0 around its check-in, sweeps and waits, in the area behind role A's)
#define BPSW_MARK() do { if ((threadIdx.x & 63) == 0 && b < 8 && fa.sw.dump && dg_pi < 40 && wave < 2) \
        fa.sw.dump[1024 + role * 640 + b * 80 + (wave ? 40 : 0) + dg_pi] = (double)(__builtin_amdgcn_s_memrealtime()); ++dg_pi; } while (0)
#else
#define BPSW_MARK() do {} while (0)
#endif
    if (role == 1) {
        // ---- role B: check in, then every gain sweep of the sample ------------------------------------------------------------------------
        BPSW_MARK();
        if (threadIdx.x == 0) {
            const long long mine = xep | 1 | ((long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 8);
            long long code = 0;
            for (int tries = 0; tries < 64; ++tries) {
                long long old = xw_load(xw);
                if ((old >> 32) == (xep >> 32)) { code = old & 0xff; break; }          // (role A was here first: solo)
                if (__hip_atomic_compare_exchange_strong(xw, &old, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { code = 1; break; }
            }
            // role A is running (it was dispatched before this workgroup): its decision comes within its initialisation
            for (int polls = 0; code == 1 && polls < (1 << 22); ++polls) {
                const long long x = xw_load(xw);
                if ((x & 0xff) != 1) code = x & 0xff; else __builtin_amdgcn_s_sleep(1);
            }
            xbc = code;
        }
        __syncthreads();
        const bool duo_b = (xbc == 4);
        __syncthreads();
        if (!duo_b) return;
        BPSW_MARK();
        SweepArgs sa = fa.sw;
        int mode = 5;                                // initialize!'s trajectory: the first step!'s gain sweep, speculatively
        for (int guard = 0; guard <= fa.max_rounds; ++guard) {
            sa.mode = mode;
            psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4g, wave);
            BPSW_MARK();
            xc_post(xw + 16, xep | ++seqB);
            // Role A's posts carry what they ask for in their two low bits (word = epoch | sequence number << 2 | request): 0 nothing (accepting
            // the candidate would end solve!: nothing consumes a gain sweep), 1 the plain gain sweep (mode 0), 2 the speculative one on the
            // candidate (mode 4), 3 the solve is over.  The request travels WITH the post, not in the sample's state words: after a post that
            // asks for nothing role A does not wait, so by the time this workgroup looks the state words may already be the next round's (a
            // late reader would sweep that round twice and run one post ahead: found by tests/test_cpu_duo_protocol.py, not by the device).
            // A post that asks for a sweep is answered before role A moves on, so what the sweep reads is stable.
            while (true) {
                const long long want = ++seqA;
                const long long got = xc_wait(xw + 8, xep | (want << 2), &xbc);
                BPSW_MARK();
                if (got == 0) return;                                                   // (timed out: role A reports it)
                const int code = (int)(got & 3);
                if (code == 3) return;                                                  // over (every post before it asked for nothing or was answered)
                if (((got - xep) >> 2) > want || code == 0) continue;                   // this post asked for nothing
                mode = (code == 1) ? 0 : 4;
                break;
            }
        }
        return;
    }
    if (leader) init_state_body(st, fa.sw.op, fa.theta_in, b);
    if (fa.init_x) {                             // initialize!'s rollout was run once for the whole batch (FusedArgs.init_*): copied by another
        if (wave == 1) copy_initial(st, fa.init_x, fa.init_u, fa.init_t, b);          // wave while the leader writes the sample's control words
    }
    BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
    if (!fa.init_x) {                            // initialize!: open-loop rollout (wave 0) + linearise (the other waves)   (ileqg.jl:214-233)
        RolloutArgs ra = fa.ro; ra.mode = 0;
        if (wave == 0) rollrec_body<0, true>(ra, b, stg, xu, &prog, epoch, d_acc);
        else rolllin_body<0, CTV, true, true>(ra, b, shxu, xu, &prog, epoch, wave - 1, 3, d_acc);
        epoch += st.N + 2;
        BPSW_MARK();
        __syncthreads();
        BPSW_MARK();
    }
    bool duo = false;
    if (fa.duo_stride > 0) {                     // is the partner there (and in this XCD)?  Decided once; the word doubles as "initialised"
        XC_DRAIN();
        __syncthreads();
        if (threadIdx.x == 0) {
            const long long myx = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);
            long long code = 2;
            for (int polls = 0; polls < 16; ++polls) {          // (a short bounded grace: the partner starts a few workgroups behind)
                long long old = xw_load(xw);
                if ((old >> 32) == (xep >> 32) && (old & 0xff) == 1) { code = (((old >> 8) & 0xff) == myx) ? 4 : 2; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (code == 2) {                                    // solo -- unless the partner checks in at this very moment
                long long old = xw_load(xw);
                while (true) {
                    if ((old >> 32) == (xep >> 32) && (old & 0xff) == 1) { code = (((old >> 8) & 0xff) == myx) ? 4 : 2; break; }
                    if (__hip_atomic_compare_exchange_strong(xw, &old, xep | 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                }
            }
            xw_store(xw, xep | code);
            if (code == 4 && fa.duo_count) atomicAdd(fa.duo_count, 1);
            xbc = code;
        }
        __syncthreads();
        duo = (xbc == 4);
        __syncthreads();
    }
    {                                            // open-loop policy evaluation (:234) || the first step!'s gain sweep on the same trajectory
        SweepArgs sa = fa.sw;
        if (duo) { sa.mode = 2; psweep_body<false, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4e, wave); }
        else if (SEQ4) {                             // W(k): one four-wave team, the evaluation and then the gain sweep (see SEQ4)
            sa.mode = 2; psweep_body<false, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4e, wave);
            __syncthreads();
            sa.mode = 5; psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4g, wave);
        }
        else if (team == 0) { sa.mode = 2; psweep_body<false, WM, false, FLYB>(sa, b, wls, &psh[0], fa.psw2e, tw); }
        else { sa.mode = 5; psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[1], fa.psw2g, tw); }
    }
    BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
    bool lost = false;                           // (a hand-over that timed out: a protocol error; the sample ends with status 8)
    if (duo) lost = xc_wait(xw + 16, xep | ++seqB, &xbc) == 0;
    BPSW_MARK();
    if (leader && !lost) commit_init_body(st, b);
    BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
    for (int guard = 0; guard < fa.max_rounds && !lost; ++guard) {
        // the round's control words in ONE batch (every wave reads the same words; rollout_active's three ride along: a dependent load is an
        // L2 round trip of ~0.7 us on the critical path)
        const int v_stat = __atomic_load_n(&st.status[b], __ATOMIC_RELAXED), v_act = __atomic_load_n(&st.ls_active[b], __ATOMIC_RELAXED);
        const int v_nom = xld(&st.slot_nom[b]), v_lsel = xld(&st.lsel[b]);
        const double v_eps = xld(&st.ls_eps[b]);
        const double v_mu = xld(&st.mu[b]);                  // (mu and the iteration count of the "would accepting end the solve" test behind the
        const int v_it = xld(&st.iter[b]);                   //  rollout: nothing writes them in between)
        if (__builtin_amdgcn_readfirstlane(v_stat) != ST_RUNNING) break;
        if (!__builtin_amdgcn_readfirstlane(v_act)) {        // step!: solve_approximate_dp! with no valid speculative sweep  (ileqg.jl:598-613)
            if (duo) {
                xc_post(xw + 8, xep | ((long long)++seqA << 2) | 1);                       // the plain gain sweep, please
                lost = xc_wait(xw + 16, xep | ++seqB, &xbc) == 0;
            } else {
                SweepArgs sa = fa.sw; sa.mode = 0;
                psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4g, wave);
            }
            BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
            continue;
        }
        if (fa.acl) {                                        // the candidate of this line-search round in deviation form (rollacl_body; switch block_acl)
            RolloutArgs ra = fa.ro; ra.mode = 1;
            const int nom_ = wave_uniform(v_nom), lsel_ = wave_uniform(v_lsel);         // (rollout_active<1>'s words: read at the loop top)
            const double eps_ = v_eps;
            const bool act = true;
            if (act) {
                if (leader) { d_acc[0] = 0ull; d_acc[1] = 0ull; lpool = 0; }
                stage_shared<4>(ra, b, nom_, lsel_, stg, wave);
            }
            BPSW_MARK();
            __syncthreads();
            BPSW_MARK();
            if (act) {
                if (!CTV && fa.prl) {                        // kappa == 0, time-invariant cost: the rollout time-parallel over the four waves
                    rollprl_body(ra, b, nom_, eps_, stg, shxu, &prl, epoch, wave, fa.prl_cut, d_acc);
                } else {
                    if (wave == 0) rollacl_body(ra, eps_, stg, aclring, xu, &pprog, &prog, epoch);
                    else if (wave == 1) rollprod_body(ra, stg, aclring, &pprog, &prog, epoch);
                    rolllin_body<1, CTV, true, true, true>(ra, b, shxu, xu, &prog, epoch, 0, 1, d_acc, stg, eps_, &lpool, wave == 0 ? 1 : 0);
                }
            }
            epoch += st.N + 2;
        } else {                                             // the candidate of this line-search round  (ileqg.jl:504-521)
            RolloutArgs ra = fa.ro; ra.mode = 1;
            if (wave == 0) rollrec_body<1, true>(ra, b, stg, xu, &prog, epoch, d_acc);
            else rolllin_body<1, CTV, true, true>(ra, b, shxu, xu, &prog, epoch, wave - 1, 3, d_acc);
            epoch += st.N + 2;
        }
        BPSW_MARK();
        if (duo) XC_DRAIN();                                 // (the candidate's trajectory is about to be handed to the partner)
        __syncthreads();
        BPSW_MARK();
        // d of the candidate, gathered by the linearise waves in LDS (d_acc): to where the accept rule reads it (ordered before select by the
        // barrier behind the sweeps); every wave forms it for the test below
        const unsigned long long da0 = d_acc[0], da1 = d_acc[1];
        const double v_dc = da1 ? NAN : sqrt(__longlong_as_double((long long)da0));
        if (threadIdx.x == 64) { st.d_c[b] = v_dc; st.flag_c[b] = 0; }
        // would accepting this candidate end solve! (:642-653)?  Then nothing consumes a speculative gain sweep: all four waves evaluate.
        const double dc = readlane_f64(v_dc, 0), mu = readlane_f64(v_mu, 0);
        const bool ends = (fa.sw.op.d > dc && mu <= fa.sw.op.mu_min) || __builtin_amdgcn_readfirstlane(v_it) == fa.sw.op.iter_max;
        if (duo) xc_post(xw + 8, xep | ((long long)++seqA << 2) | (ends ? 0 : 2));      // the speculative gain sweep on this candidate -- unless accepting it ends the solve
        {
            SweepArgs sa = fa.sw;
            if (ends || duo) { sa.mode = 1; psweep_body<false, WM, true, FLYB>(sa, b, wls, &psh[2], fa.psw4e, wave); }
            else if (SEQ4) {
                sa.mode = 1; psweep_body<false, WM, true, FLYB>(sa, b, wls, &psh[2], fa.psw4e, wave);
                __syncthreads();
                sa.mode = 4; psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[2], fa.psw4g, wave);
            }
            else if (team == 0) { sa.mode = 1; psweep_body<false, WM, true, FLYB>(sa, b, wls, &psh[0], fa.psw2e, tw); }
            else { sa.mode = 4; psweep_body<true, WM, false, FLYB>(sa, b, wls, &psh[1], fa.psw2g, tw); }
        }
        BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
        if (duo && !ends) lost = xc_wait(xw + 16, xep | ++seqB, &xbc) == 0;
        BPSW_MARK();
        if (leader && !lost) ls_select_body(st, fa.sw.op, b, nullptr);
        BPSW_MARK();
    __syncthreads();
    BPSW_MARK();
    }
    BPSW_MARK();
    if (duo) xc_post(xw + 8, xep | ((long long)++seqA << 2) | 3);       // over (always: the partner must not be left waiting)
    __syncthreads();
    BPSW_MARK();
    if (leader) {
        if (lost) { st.status[b] = 8; st.value[b] = INFINITY; }
        gather_body(st, b, fa.out_value, fa.out_status, fa.out_iters, fa.out_ls, fa.out_cost, fa.kl_bound);
    }
}

#undef BPSW_MARK
void launch_solve_block_psw_tv(const FusedArgs &fa, const dim3 grid, hipStream_t s);
#if RAT_PART & PART_BPSW
bool solve_block_psw_supported(const FusedArgs &fa) {
    return fa.sw.st.E == 1 && fa.sw.pb.model == 1 && fa.sw.st.N <= ROLLIN_NST && fa.sw.st.N >= 8;
}
void launch_solve_block_psw(const FusedArgs &fa, hipStream_t s) {
    if (fa.sw.st.B <= 0) return;
    // duo (two workgroups per sample, FusedArgs.duo_stride): groups of 16 blocks = 8 samples x {role A, role B}
    const dim3 grid(fa.duo_stride > 0 ? 2 * fa.duo_stride : fa.sw.st.B), block(256);
    if (fa.sw.pb.W_tv) { launch_solve_block_psw_tv(fa, grid, s); return; }
#define BPSW_LAUNCH(C) do { if (fa.sw.pb.W_diag) hipLaunchKernelGGL((solve_block_psw_kernel<C, 2>), grid, block, 0, s, fa); \
                            else hipLaunchKernelGGL((solve_block_psw_kernel<C, 0>), grid, block, 0, s, fa); } while (0)
    if (fa.sw.pb.cost_tv) BPSW_LAUNCH(true); else BPSW_LAUNCH(false);
#undef BPSW_LAUNCH
}
#endif  // PART_BPSW
// Time-varying W(k): the same kernel's <.., W_tv> instantiations, in a translation-unit part of their own that is built WITHOUT
// -amdgpu-mfma-vgpr-form.  With that flag clang 22 miscompiles them: the gain sweeps of a workgroup that also runs evaluations come out wrong
// (negative pivots behind a hop: every line-search candidate then fails), while the identical source is right without the flag, right with it
// when the gain sweeps have a workgroup to themselves (role B of the two-workgroup launch), and right in psweep_kernel.  Round 5 had this down
// as "the <fly, W_tv> instantiation is not understood"; tools/wtv_probe.py is the reproducer.  The flag's pass is the one that crashes on
// kernels that spill (wide.hip); the instantiations built with it are the ones the parity suite, soaks and race hunts hold to the oracle.
#if RAT_PART & PART_BPSW1
void launch_solve_block_psw_tv(const FusedArgs &fa, const dim3 grid, hipStream_t s) {
    if (fa.sw.pb.cost_tv) hipLaunchKernelGGL((solve_block_psw_kernel<true, 1>), grid, dim3(256), 0, s, fa);
    else hipLaunchKernelGGL((solve_block_psw_kernel<false, 1>), grid, dim3(256), 0, s, fa);
}
#endif  // PART_BPSW1

template <int NW, bool GW>
static void launch_solve_block_n(const FusedArgs &fa, hipStream_t s) {
    const dim3 grid(fa.sw.st.B), block(64 * NW);
    const int wm = fa.sw.pb.W_tv ? 1 : (fa.sw.pb.W_diag ? 2 : 0);
    const bool stg = fa.sw.pb.model == 1 && fa.sw.st.N <= ROLLIN_NST;
#define BLOCK_LAUNCH(M, C, W, S) do { \
        if (NW == 2 && fa.census) hipLaunchKernelGGL((solve_block_kernel<M, C, W, NW, GW, S, NW == 2>), grid, dim3(256), 0, s, fa); \
        else hipLaunchKernelGGL((solve_block_kernel<M, C, W, NW, GW, S, false>), grid, block, 0, s, fa); } while (0)
    if (fa.sw.pb.model == 1) {
        if (stg) {
#define BLOCK_W(M, C, S) do { if (wm == 1) BLOCK_LAUNCH(M, C, 1, S); else if (wm == 2) BLOCK_LAUNCH(M, C, 2, S); else BLOCK_LAUNCH(M, C, 0, S); } while (0)
            if (fa.sw.pb.cost_tv) BLOCK_W(1, true, true); else BLOCK_W(1, false, true);
        } else {
            if (fa.sw.pb.cost_tv) BLOCK_W(1, true, false); else BLOCK_W(1, false, false);
        }
    } else {
        BLOCK_W(2, false, false);
    }
#undef BLOCK_W
#undef BLOCK_LAUNCH
}

// E = st.E speculative candidates per sample: E + 1 waves (the last one runs the gain sweeps) for E <= 7; E waves for E = 8, the last one
// doubling as gain wave and lazy evaluator of candidate 7 (LAZY in solve_block_kernel).
// One translation-unit part per geometry (the Makefile compiles this file once per RAT_PART: the instantiations of one geometry are a
// minute of compile time each).
void launch_solve_block_e1(const FusedArgs &fa, hipStream_t s);
void launch_solve_block_e2(const FusedArgs &fa, hipStream_t s);
void launch_solve_block_e4(const FusedArgs &fa, hipStream_t s);
void launch_solve_block_e8(const FusedArgs &fa, hipStream_t s);
#if RAT_PART & PART_BLOCK2
void launch_solve_block_e1(const FusedArgs &fa, hipStream_t s) { launch_solve_block_n<2, true>(fa, s); }
#endif
#if RAT_PART & PART_BLOCK3
void launch_solve_block_e2(const FusedArgs &fa, hipStream_t s) { launch_solve_block_n<3, true>(fa, s); }
#endif
#if RAT_PART & PART_BLOCK5
void launch_solve_block_e4(const FusedArgs &fa, hipStream_t s) { launch_solve_block_n<5, true>(fa, s); }
#endif
#if RAT_PART & PART_BLOCK8
void launch_solve_block_e8(const FusedArgs &fa, hipStream_t s) { launch_solve_block_n<8, true>(fa, s); }
#endif
#if RAT_PART & PART_ROLL
bool solve_block_supported(int E) { return E == 1 || E == 2 || E == 4 || E == 8; }
void launch_solve_block(const FusedArgs &fa, hipStream_t s) {
    if (fa.sw.st.B <= 0) return;
    switch (fa.sw.st.E) {
        case 1: launch_solve_block_e1(fa, s); break;
        case 2: launch_solve_block_e2(fa, s); break;
        case 4: launch_solve_block_e4(fa, s); break;
        case 8: launch_solve_block_e8(fa, s); break;
        default: break;
    }
}

void launch_init_state(const StateDev &st, const OptsDev &op, const double *theta_dev, hipStream_t s) {
    const int n = st.B > 2 * CTR_RING ? st.B : 2 * CTR_RING;
    hipLaunchKernelGGL(init_state_kernel, dim3((n + 255) / 256), dim3(256), 0, s, st, op, theta_dev);
}
// round-based tile-free path: the samples' nominal slots <- initialize!'s shared trajectory (one wavefront per sample)
__global__ __launch_bounds__(64) void copy_initial_kernel(StateDev st, const double *init_x, const double *init_u, const double *init_t) {
    copy_initial(st, init_x, init_u, init_t, blockIdx.x);
}
void launch_copy_initial(const StateDev &st, const double *init_x, const double *init_u, const double *init_t, hipStream_t s) {
    if (st.B <= 0) return;
    hipLaunchKernelGGL(copy_initial_kernel, dim3(st.B), dim3(64), 0, s, st, init_x, init_u, init_t);
}
void launch_ls_select(const StateDev &st, const OptsDev &op, int slot, hipStream_t s) {
    hipLaunchKernelGGL(ls_select_kernel, dim3((st.B + 255) / 256), dim3(256), 0, s, st, op, slot);
}

#endif  // PART_ROLL
// gather the outputs of a batch: value (Inf for failures), status, iters, ls_evals
// the outputs of sample b: value (Inf for failures), status, iters, ls_evals, cost = value + kl / theta (cross_entropy...jl:193)
__device__ __forceinline__ void gather_body(const StateDev &st, const int b, double *value, int *status, int *iters, int *ls_evals,
                                            double *cost, double kl_bound) {
    const int s = xld(&st.status[b]);
    const double v = (s == 0 || s == 3) ? xld(&st.value[b]) : INFINITY;
    if (value) value[b] = v;
    if (status) status[b] = s;
    if (iters) iters[b] = xld(&st.iter[b]);
    if (ls_evals) ls_evals[b] = xld(&st.n_ls[b]);
    if (cost) cost[b] = v + kl_bound / xld(&st.theta[b]);
}
#if RAT_PART & PART_ROLL
__global__ void gather_kernel(StateDev st, double *value, int *status, int *iters, int *ls_evals, double *cost, double kl_bound) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= st.B) return;
    gather_body(st, b, value, status, iters, ls_evals, cost, kl_bound);
}
// The result of ONE solve (sample 0 of the handle: rat_ileqg_solve, the final solve of rat_ce_solve) into one destination -- pinned host
// memory the device writes in place: the accepted trajectory (slot slot_nom), the committed gains (half lsel) and the scalars.  Nine
// device-to-host copies (each a copy kernel of ~5 us plus its gap in the stream) used to follow the solve; now one launch does.
__global__ __launch_bounds__(256) void pack_solution_kernel(StateDev st, int b, double *dst_x, double *dst_u, double *dst_L, double *dst_s) {
    const int nom = st.slot_nom[b], lsel = st.lsel[b];
    const long slot = (long)b * (st.E + 1) + nom;
    const double *__restrict__ xs = st.xs + slot * st.x_stride, *__restrict__ us = st.us + slot * st.u_stride;
    const double *__restrict__ Ls = st.L + (long)lsel * st.l_half + (long)b * st.N * LSTR;
    const int nx = (int)st.x_stride, nu = (int)st.u_stride, nL = st.N * LSTR;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, str = gridDim.x * blockDim.x;
    if (dst_x) for (int i = i0; i < nx; i += str) dst_x[i] = xs[i];
    if (dst_u) for (int i = i0; i < nu; i += str) dst_u[i] = us[i];
    if (dst_L) for (int i = i0; i < nL; i += str) dst_L[i] = Ls[i];
    if (i0 == 0) {
        dst_s[0] = st.value[b]; dst_s[1] = (double)st.status[b]; dst_s[2] = (double)st.iter[b]; dst_s[3] = (double)st.hist_n[b];
        dst_s[4] = (double)nom; dst_s[5] = (double)lsel;
    }
}
void launch_pack_solution(const StateDev &st, int b, double *dst_x, double *dst_u, double *dst_L, double *dst_s, hipStream_t s) {
    hipLaunchKernelGGL(pack_solution_kernel, dim3(4), dim3(256), 0, s, st, b, dst_x, dst_u, dst_L, dst_s);
}
void launch_gather(const StateDev &st, double *value, int *status, int *iters, int *ls_evals, double *cost, double kl_bound, hipStream_t s) {
    hipLaunchKernelGGL(gather_kernel, dim3((st.B + 255) / 256), dim3(256), 0, s, st, value, status, iters, ls_evals, cost, kl_bound);
}

#endif  // PART_ROLL

#if RAT_PART & PART_MISC
// =====================================================================================================
// PETS (pets.jl:76-157): stochastic forward rollouts with running cost, 4 trajectories per wavefront (16-lane rows).
// Noise comes from an injected stream (parity with the oracle, serial semantics) or from Philox4x32-10 on the device.
// =====================================================================================================
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

__global__ __launch_bounds__(64) void pets_rollout_kernel(PetsArgs a) {
    const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
    const GenDev &g = a.g;
    const long ntraj = a.S * a.K;
    const long tj = (long)blockIdx.x * 4 + row;
    const bool live = tj < ntraj;
    const long tjg = tj + a.traj0;         // global trajectory index: keys the device generator, so that a shard of the control samples draws
                                           // the noise it would draw as part of the whole batch (results do not depend on the sharding)
    const long ii = live ? tj / a.K : 0;
    __shared__ double shxu[4][16];
    __shared__ double shz[4][16];
    const int N = g.N, n = g.n;
    const int jx = (j < 12) ? j : 11;
    double zr[16], crow[16], nrow[12], trow[12];
#pragma unroll
    for (int q = 0; q < 16; ++q) zr[q] = g.Zt[jx * 16 + q];
#pragma unroll
    for (int q = 0; q < 12; ++q) { nrow[q] = g.nchol[jx * 16 + q]; trow[q] = g.tchol2 ? g.tchol2[jx * 16 + q] : 0.0; }
    const double nmean = g.nmean[jx], tmean = g.tmean2 ? g.tmean2[jx] : 0.0;
    double x = (j < 12) ? a.x0[j] : 0.0;
    double cacc = 0.0, znext = 0.0;
    const double *__restrict__ uc = a.controls + ii * N * USTR;
    const bool need_sel = a.use_true && g.tw2 > 0.0;          // the mixture selector is only drawn when a second component exists
    double lin = g.lin[j], q0t = g.q0[0];
#pragma unroll
    for (int q = 0; q < 16; ++q) crow[q] = g.Ctab[j * 16 + q];   // time-invariant cost tables stay in registers
    for (int t = 0; t < N; ++t) {
        if (g.cost_tv) {
#pragma unroll
            for (int q = 0; q < 16; ++q) crow[q] = g.Ctab[(long)t * 256 + j * 16 + q];
            lin = g.lin[(long)t * 16 + j];
            q0t = g.q0[t];
        }
        const double u = uc[(long)t * USTR + (j & 3)];
        // draws of this (trajectory, step): zn (per state lane) and the mixture selector
        double z = 0.0, zsel = 1.0;
        if (a.zn) {
            if (j < n && live) z = a.zn[((tj * N + t) * (long)n) + j];
            if (a.zu && live) zsel = a.zu[tj * N + t];
        } else {
            // one Philox block and one Box-Muller transform serve two consecutive steps (both outputs of the transform are used)
            if ((t & 1) == 0) {
                unsigned r[4];
                philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)(t >> 1), (unsigned)j, (unsigned)a.seed, (unsigned)(a.seed >> 32), r);
                const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
                if (g.noise_kind == 1 && !need_sel) { z = u1; znext = u2; }
                else {
                    ratn_box_muller(u1, u2, &z, &znext);     // (rat_normal.h: the transform written out for its argument ranges)
                }
            } else z = znext;
            if (need_sel) {
                unsigned rs[4];
                philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)t, 0xFFFFu, (unsigned)a.seed, (unsigned)(a.seed >> 32), rs);
                zsel = u01(rs[0], rs[1]);
            }
        }
        if (j < 12) shxu[row][j] = x;
        if (j < 4) shxu[row][12 + j] = u;
        shz[row][j] = (j < n) ? z : 0.0;
        WAVE_SYNC();
        // stage cost c(t, x, u) = 1/2 xu' C xu + lin' xu + q0 + l1u sum|u|      (pets.jl:143)
        double acc = 0.0, dyn = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc = fma(crow[q], shxu[row][q], acc); dyn = fma(zr[q], shxu[row][q], dyn); }
        const double xuj = (j < 12) ? x : u;
        double part = xuj * (0.5 * acc + lin) + ((j >= 12 && j - 12 < g.m) ? g.l1u * fabs(u) : 0.0);
        part = row_sum16(part);
        cacc += part + q0t;
        // stochastic transition x <- f_stochastic(x, u, rng, use_true_model)     (pets.jl:144)
        if (g.kappa != 0.0) dyn += g.kappa * (x * x * x);
        double w;
        const bool second = a.use_true && (zsel < g.tw2);
        if (second || g.noise_kind == 0) {
            double nz = 0.0;
#pragma unroll
            for (int q = 0; q < 12; ++q) nz = fma(second ? trow[q] : nrow[q], shz[row][q], nz);
            w = (second ? tmean : nmean) + nz;
        } else {
            w = g.nlo + (g.nhi - g.nlo) * z;
        }
        x = (j < n) ? dyn + w : 0.0;
        WAVE_SYNC();
    }
    // terminal cost h(x_N)     (pets.jl:147)
    if (j < 12) shxu[row][j] = x;
    WAVE_SYNC();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 12; ++q) acc = fma(g.Qf[jx * 12 + q], shxu[row][q], acc);
    double part = (j < 12) ? x * (0.5 * acc + g.qvf[jx]) : 0.0;
    part = row_sum16(part);
    if (live && j == 0) a.traj_cost[tj] = cacc + part + g.q0f;
}

// =====================================================================================================
// noisy_rollout_kernel: Monte-Carlo rollouts under process noise, x_{k+1} = f(x_k, u_k) + w_k, w_k ~ N(0, W(k)), open loop or under
// the affine policy u_k = l_k + L_k (x_k - xbar_k)   (simulate_dynamics with rng, ileqg.jl:44-55 / :94-109), with the realised
// cost of every rollout (integrate_cost, :115-124).  Four rollouts per wavefront (one per 16-lane row), lane j < 12 owns state j.
// w_k = chol_lower(W(k)) z_k with z_k from an injected standard-normal stream (parity with the oracle) or Philox4x32-10.
// =====================================================================================================
__global__ __launch_bounds__(64) void noisy_rollout_kernel(NoisyArgs a) {
    const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
    const ProblemDev &pb = a.pb;
    const long k = (long)blockIdx.x * 4 + row;
    const bool live = k < a.K;
    const int N = pb.N, n = pb.n;
    __shared__ double shdx[4][12], shxu[4][16], shz[4][16];
    const int jx = (j < 12) ? j : 11, ju = j & 3;
    const bool lq = (pb.model == 1);
    double zr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) zr[q] = lq ? pb.Zt[jx * 16 + q] : 0.0;
    double x = (j < 12) ? a.xnom[j] : 0.0;
    double cacc = 0.0, znext = 0.0;
    int dom = 0;
    for (int t = 0; t < N; ++t) {
        const int kc = pb.cost_tv ? t : 0, kw = pb.W_tv ? t : 0;
        if (live && j < 12 && a.x_out) a.x_out[(k * (N + 1) + t) * XSTR + j] = x;
        double u = a.l[(long)t * USTR + ju];
        if (a.L) {                                                   // L_t (x_t - xbar_t)   (:104)
            if (j < 12) shdx[row][j] = x - a.xnom[(long)t * XSTR + j];
            WAVE_SYNC();
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = fma(a.L[(long)t * LSTR + ju * 12 + q], shdx[row][q], acc);
            u += acc;
        }
        if (live && j < 4 && a.u_out) a.u_out[(k * N + t) * USTR + j] = u;
        double z = 0.0;
        if (a.z) { if (live && j < n) z = a.z[(k * N + t) * (long)n + j]; }
        else {
            if ((t & 1) == 0) {                                      // both outputs of one Box-Muller transform: steps t and t + 1
                unsigned r[4];
                philox4x32_10((unsigned)k, (unsigned)(k >> 32), (unsigned)(t >> 1), (unsigned)j, (unsigned)a.seed, (unsigned)(a.seed >> 32), r);
                ratn_box_muller(u01(r[0], r[1]), u01(r[2], r[3]), &z, &znext);
            } else z = znext;
        }
        if (j < 12) shxu[row][j] = x;
        if (j < 4) shxu[row][12 + j] = u;
        shz[row][j] = (j < n) ? z : 0.0;
        WAVE_SYNC();
        double xn = 0.0, part = 0.0, q0 = 0.0;
        if (lq) {
            double acc = 0.0, dyn = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc = fma(pb.Ctab[(long)kc * 256 + j * 16 + q], shxu[row][q], acc);
                dyn = fma(zr[q], shxu[row][q], dyn);
            }
            if (pb.kappa != 0.0) dyn += pb.kappa * (x * x * x);
            xn = dyn;
            const double xuj = (j < 12) ? x : u;
            part = xuj * (0.5 * acc + pb.lin[(long)kc * 16 + j]);     // c = 1/2 xu' C xu + lin' xu + q0
            q0 = pb.q0[kc];
        } else {
            if (j < n) {
                xn = powchk(x, pb.pl_a, dom) + powchk(shxu[row][12 + ju], pb.pl_b, dom);
                part = pb.pl_cx * powchk(x, pb.pl_p, dom) + pb.pl_cu * powchk(shxu[row][12 + ju], pb.pl_pu, dom);
            }
        }
        cacc += row_sum16(part) + q0;
        double w = 0.0;                                               // w_k = chol_lower(W(k)) z_k
#pragma unroll
        for (int q = 0; q < 12; ++q) w = fma(a.Wchol[(long)kw * 192 + jx * 16 + q], shz[row][q], w);
        x = (j < n) ? xn + w : 0.0;
        WAVE_SYNC();
    }
    if (live && j < 12 && a.x_out) a.x_out[(k * (N + 1) + N) * XSTR + j] = x;
    // terminal cost h(x_N)
    if (j < 12) shxu[row][j] = x;
    WAVE_SYNC();
    double hcost;
    if (lq) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc = fma(pb.Qf[jx * 12 + q], shxu[row][q], acc);
        hcost = row_sum16((j < 12) ? x * (0.5 * acc + pb.qvf[jx]) : 0.0) + pb.q0f;
    } else {
        hcost = pb.pl_h;
    }
    const unsigned long long bal = __ballot(dom != 0);
    if (live && j == 0) {
        if (a.cost) a.cost[k] = cacc + hcost;
        if (a.dom) a.dom[k] = ((bal >> (row * 16)) & 0xFFFFull) != 0;
    }
}

void launch_noisy_rollout(const NoisyArgs &a, hipStream_t s) {
    if (a.K <= 0) return;
    hipLaunchKernelGGL(noisy_rollout_kernel, dim3((unsigned)((a.K + 3) / 4)), dim3(64), 0, s, a);
}

// mean over the K rollouts of each control sample (pets.jl:150): one wavefront per sample, lane l sums rollouts l, l + 64, ... in
// order, then the 64 partial sums in a fixed tree -- deterministic, independent of the launch geometry (a thread per sample summing K
// values one after the other took 160 us at K = 1000)
// =====================================================================================================
// pets_rollout16_kernel (round 4): SIXTEEN stochastic rollouts per wavefront -- trajectory j is column j of every MFMA's B operand, the
// state lives in B-form (register r of lane (g, j): component 4 r + g of trajectory j), as in rollin_multi_kernel.  Per step
//     C [x; u]           4 MFMAs   -> stage cost  [x;u]' (1/2 C [x;u] + lin) + q0 + l1u |u|_1     (pets.jl:143), summed over a column's four rows
//     [A | B] [x; u]     4 MFMAs   -> f(x, u) (+ kappa x^3 per lane)
//     chol z             3 MFMAs   -> w = mean + chol z   (3 more for the second mixture component of the true model; none for uniform noise)
// against 44 multiply-adds and 44 LDS reads per lane and step for FOUR trajectories in pets_rollout_kernel.  Every lane draws the three
// normals of ITS components (4 r + g, r < 3) of ITS trajectory: the Philox key is the one of pets_rollout_kernel -- (trajectory, step pair,
// component) -- so the device generator hands the SAME normals to the same (trajectory, step, component); only now all 64 lanes of the wave
// generate useful draws (12 of every 16 before).  Sums run in the MFMA's order: costs agree with the 4-per-wave kernel and the oracle to
// rounding (tests/test_gpu_pets.py: 1e-11 on injected noise).
// =====================================================================================================
__global__ __launch_bounds__(64) void pets_rollout16_kernel(PetsArgs a) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const GenDev &gd = a.g;
    const long ntraj = a.S * a.K;
    const long tj = (long)blockIdx.x * 16 + j;
    const bool live = tj < ntraj;
    const long tjc = live ? tj : 0;
    const long tjg = tj + a.traj0;
    const long ii = tjc / a.K;
    const int N = gd.N, n = gd.n;
    const int jr = (j < 12) ? j : 11;
    const double mq = (j < 12) ? 1.0 : 0.0;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    // A operands: lane (g, j) of slice s holds M[row j][4 s + g]
    double zA[4], cA[4], nA[3], tA[3], qA[3];
#pragma unroll
    for (int s = 0; s < 4; ++s) { zA[s] = gd.Zt[jr * 16 + 4 * s + g] * mq; cA[s] = gd.Ctab[j * 16 + 4 * s + g]; }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        nA[s] = gd.nchol[jr * 16 + 4 * s + g] * mq;
        tA[s] = gd.tchol2 ? gd.tchol2[jr * 16 + 4 * s + g] * mq : 0.0;
        qA[s] = gd.Qf[jr * 12 + 4 * s + g] * mq;
    }
    // per-component constants in B-form
    double linb[4], nmb[3], tmb[3], qvb[3], cmask[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) linb[r] = gd.lin[4 * r + g];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        nmb[r] = gd.nmean[4 * r + g]; tmb[r] = gd.tmean2 ? gd.tmean2[4 * r + g] : 0.0; qvb[r] = gd.qvf[4 * r + g];
        cmask[r] = (4 * r + g < n) ? 1.0 : 0.0;
    }
    double q0t = gd.q0[0];
    const double l1u = (g < gd.m) ? gd.l1u : 0.0;
    const bool need_sel = a.use_true && gd.tw2 > 0.0;          // the mixture selector is only drawn when a second component exists
    const bool gauss = gd.noise_kind == 0;
    double xs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) xs[r] = a.x0[4 * r + g];
    const double *__restrict__ uc = a.controls + ii * N * USTR + g;
    double cacc = 0.0;
    double zn1[3] = {0.0, 0.0, 0.0};
    double unext = uc[0];
    for (int t = 0; t < N; ++t) {
        if (gd.cost_tv) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { cA[s] = gd.Ctab[(long)t * 256 + j * 16 + 4 * s + g]; linb[s] = gd.lin[(long)t * 16 + 4 * s + g]; }
            q0t = gd.q0[t];
        }
        const double u = unext;
        unext = uc[(long)((t + 1 < N) ? t + 1 : t) * USTR];
        // draws of this (trajectory, step): z (the lane's three components) and the mixture selector
        double z[3], zsel = 1.0;
        if (a.zn) {
#pragma unroll
            for (int r = 0; r < 3; ++r) z[r] = (4 * r + g < n && live) ? a.zn[((tjc * N + t) * (long)n) + 4 * r + g] : 0.0;
            if (a.zu && live) zsel = a.zu[tjc * N + t];
        } else {
            if ((t & 1) == 0) {                                // one Philox block and one Box-Muller transform serve two consecutive steps
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    unsigned rr[4];
                    philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)(t >> 1), (unsigned)(4 * r + g), (unsigned)a.seed, (unsigned)(a.seed >> 32), rr);
                    const double u1 = u01(rr[0], rr[1]), u2 = u01(rr[2], rr[3]);
                    if (!gauss && !need_sel) { z[r] = u1; zn1[r] = u2; }
                    else {
                        ratn_box_muller(u1, u2, &z[r], &zn1[r]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) z[r] = zn1[r];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) z[r] *= cmask[r];
            if (need_sel) {
                unsigned rs[4];
                philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)t, 0xFFFFu, (unsigned)a.seed, (unsigned)(a.seed >> 32), rs);
                zsel = u01(rs[0], rs[1]);
            }
        }
        // stage cost
        d4 cx = MFMA(cA[0], xs[0], zero4);
        cx = MFMA(cA[1], xs[1], cx);
        cx = MFMA(cA[2], xs[2], cx);
        cx = MFMA(cA[3], u, cx);
        // dynamics
        d4 xa = MFMA(zA[0], xs[0], zero4);
        xa = MFMA(zA[1], xs[1], xa);
        xa = MFMA(zA[2], xs[2], xa);
        xa = MFMA(zA[3], u, xa);
        double part = ((xs[0] * (0.5 * cx[0] + linb[0]) + xs[1] * (0.5 * cx[1] + linb[1])) + xs[2] * (0.5 * cx[2] + linb[2])) +
                      (u * (0.5 * cx[3] + linb[3]) + l1u * fabs(u));
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        cacc += part + q0t;
        // stochastic transition x <- f_stochastic(x, u, rng, use_true_model)     (pets.jl:144)
        double w[3];
        const bool second = a.use_true && (zsel < gd.tw2);
        if (gauss) {
            d4 wz = MFMA(nA[0], z[0], zero4);
            wz = MFMA(nA[1], z[1], wz);
            wz = MFMA(nA[2], z[2], wz);
#pragma unroll
            for (int r = 0; r < 3; ++r) w[r] = nmb[r] + wz[r];
            if (need_sel) {
                d4 w2 = MFMA(tA[0], z[0], zero4);
                w2 = MFMA(tA[1], z[1], w2);
                w2 = MFMA(tA[2], z[2], w2);
#pragma unroll
                for (int r = 0; r < 3; ++r) w[r] = second ? tmb[r] + w2[r] : w[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) w[r] = gd.nlo + (gd.nhi - gd.nlo) * z[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double dyn = xa[r];
            if (gd.kappa != 0.0) dyn += gd.kappa * (xs[r] * xs[r] * xs[r]);
            xs[r] = (dyn + w[r]) * cmask[r];
        }
    }
    // terminal cost h(x_N)     (pets.jl:147)
    d4 qx = MFMA(qA[0], xs[0], zero4);
    qx = MFMA(qA[1], xs[1], qx);
    qx = MFMA(qA[2], xs[2], qx);
    double part = (xs[0] * (0.5 * qx[0] + qvb[0]) + xs[1] * (0.5 * qx[1] + qvb[1])) + xs[2] * (0.5 * qx[2] + qvb[2]);
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (live && g == 0) a.traj_cost[tj] = cacc + part + gd.q0f;
}

// pets_rollout16s_kernel: the same sixteen-column recursion with the noise generation moved OFF the recursion's wavefront.  A launch of
// <= 1024 wavefronts of pets_rollout16_kernel (16k trajectories: BASELINE config 5 has 10k) leaves most of the chip's 1024 SIMDs idle and
// its time is ONE wavefront's latency -- 30 steps of (Philox + Box-Muller ~2000 cycles, then eleven MFMAs ~800 cycles).  Here a workgroup
// is four wavefronts: waves 1..3 draw component register r = wave - 1 of every lane for one step PAIR (one Philox block and one Box-Muller
// transform yield the normals of steps 2p and 2p + 1) into a double-buffered LDS slab while wave 0 consumes the previous pair; one
// __syncthreads() per pair hands over.  Lane (g, j) of generator wave r computes exactly what lane (g, j) computed for register r before,
// and wave 0's arithmetic is unchanged: the costs are bit-identical to pets_rollout16_kernel's (tests/test_gpu_pets.py).
__global__ __launch_bounds__(256) void pets_rollout16s_kernel(PetsArgs a) {
    __shared__ double zbuf[2][2][3][64];                       // [pair parity][step of the pair][register r][lane]
    __shared__ double sbuf[2][2][16];                          // mixture selector of trajectory column j
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, j = l & 15;
    const GenDev &gd = a.g;
    const long ntraj = a.S * a.K;
    const long tj = (long)blockIdx.x * 16 + j;
    const bool live = tj < ntraj;
    const long tjc = live ? tj : 0;
    const long tjg = tj + a.traj0;
    const int N = gd.N, n = gd.n;
    const int npair = (N + 1) >> 1;
    const bool need_sel = a.use_true && gd.tw2 > 0.0;
    const bool gauss = gd.noise_kind == 0;
    if (wv > 0) {
        const int r = wv - 1, comp = 4 * r + g;
        const double cm = (comp < n) ? 1.0 : 0.0;
        for (int p = 0; p <= npair; ++p) {
            if (p < npair) {
                const int t = 2 * p;
                const bool two = t + 1 < N;
                double z0, z1 = 0.0, s0 = 1.0, s1 = 1.0;
                if (a.zn) {
                    const bool on = comp < n && live;
                    z0 = on ? a.zn[((tjc * N + t) * (long)n) + comp] : 0.0;
                    if (two) z1 = on ? a.zn[((tjc * N + t + 1) * (long)n) + comp] : 0.0;
                    if (r == 0 && a.zu && live) { s0 = a.zu[tjc * N + t]; if (two) s1 = a.zu[tjc * N + t + 1]; }
                } else {
                    unsigned rr[4];
                    philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)p, (unsigned)comp, (unsigned)a.seed, (unsigned)(a.seed >> 32), rr);
                    const double u1 = u01(rr[0], rr[1]), u2 = u01(rr[2], rr[3]);
                    if (!gauss && !need_sel) { z0 = u1; z1 = u2; }
                    else {
                        ratn_box_muller(u1, u2, &z0, &z1);
                    }
                    z0 *= cm; z1 *= cm;
                    if (r == 0 && need_sel) {
                        unsigned rs[4];
                        philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)t, 0xFFFFu, (unsigned)a.seed, (unsigned)(a.seed >> 32), rs);
                        s0 = u01(rs[0], rs[1]);
                        if (two) {
                            philox4x32_10((unsigned)tjg, (unsigned)(tjg >> 32), (unsigned)(t + 1), 0xFFFFu, (unsigned)a.seed, (unsigned)(a.seed >> 32), rs);
                            s1 = u01(rs[0], rs[1]);
                        }
                    }
                }
                zbuf[p & 1][0][r][l] = z0; zbuf[p & 1][1][r][l] = z1;
                if (r == 0 && g == 0) { sbuf[p & 1][0][j] = s0; sbuf[p & 1][1][j] = s1; }
            }
            __syncthreads();
        }
        return;
    }
    const long ii = tjc / a.K;
    const int jr = (j < 12) ? j : 11;
    const double mq = (j < 12) ? 1.0 : 0.0;
    const d4 zero4 = {0.0, 0.0, 0.0, 0.0};
    double zA[4], cA[4], nA[3], tA[3], qA[3];
#pragma unroll
    for (int s = 0; s < 4; ++s) { zA[s] = gd.Zt[jr * 16 + 4 * s + g] * mq; cA[s] = gd.Ctab[j * 16 + 4 * s + g]; }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        nA[s] = gd.nchol[jr * 16 + 4 * s + g] * mq;
        tA[s] = gd.tchol2 ? gd.tchol2[jr * 16 + 4 * s + g] * mq : 0.0;
        qA[s] = gd.Qf[jr * 12 + 4 * s + g] * mq;
    }
    double linb[4], nmb[3], tmb[3], qvb[3], cmask[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) linb[r] = gd.lin[4 * r + g];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        nmb[r] = gd.nmean[4 * r + g]; tmb[r] = gd.tmean2 ? gd.tmean2[4 * r + g] : 0.0; qvb[r] = gd.qvf[4 * r + g];
        cmask[r] = (4 * r + g < n) ? 1.0 : 0.0;
    }
    double q0t = gd.q0[0];
    const double l1u = (g < gd.m) ? gd.l1u : 0.0;
    double xs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) xs[r] = a.x0[4 * r + g];
    const double *__restrict__ uc = a.controls + ii * N * USTR + g;
    double cacc = 0.0;
    double unext = uc[0];
    __syncthreads();                                           // pair 0 is in the slab
    for (int p = 0; p < npair; ++p) {
        for (int hh = 0; hh < 2; ++hh) {
            const int t = 2 * p + hh;
            if (t >= N) break;
            if (gd.cost_tv) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { cA[s] = gd.Ctab[(long)t * 256 + j * 16 + 4 * s + g]; linb[s] = gd.lin[(long)t * 16 + 4 * s + g]; }
                q0t = gd.q0[t];
            }
            const double u = unext;
            unext = uc[(long)((t + 1 < N) ? t + 1 : t) * USTR];
            double z[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) z[r] = zbuf[p & 1][hh][r][l];
            const double zsel = sbuf[p & 1][hh][j];
            d4 cx = MFMA(cA[0], xs[0], zero4);
            cx = MFMA(cA[1], xs[1], cx);
            cx = MFMA(cA[2], xs[2], cx);
            cx = MFMA(cA[3], u, cx);
            d4 xa = MFMA(zA[0], xs[0], zero4);
            xa = MFMA(zA[1], xs[1], xa);
            xa = MFMA(zA[2], xs[2], xa);
            xa = MFMA(zA[3], u, xa);
            double part = ((xs[0] * (0.5 * cx[0] + linb[0]) + xs[1] * (0.5 * cx[1] + linb[1])) + xs[2] * (0.5 * cx[2] + linb[2])) +
                          (u * (0.5 * cx[3] + linb[3]) + l1u * fabs(u));
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            cacc += part + q0t;
            double w[3];
            const bool second = a.use_true && (zsel < gd.tw2);
            if (gauss) {
                d4 wz = MFMA(nA[0], z[0], zero4);
                wz = MFMA(nA[1], z[1], wz);
                wz = MFMA(nA[2], z[2], wz);
#pragma unroll
                for (int r = 0; r < 3; ++r) w[r] = nmb[r] + wz[r];
                if (need_sel) {
                    d4 w2 = MFMA(tA[0], z[0], zero4);
                    w2 = MFMA(tA[1], z[1], w2);
                    w2 = MFMA(tA[2], z[2], w2);
#pragma unroll
                    for (int r = 0; r < 3; ++r) w[r] = second ? tmb[r] + w2[r] : w[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) w[r] = gd.nlo + (gd.nhi - gd.nlo) * z[r];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double dyn = xa[r];
                if (gd.kappa != 0.0) dyn += gd.kappa * (xs[r] * xs[r] * xs[r]);
                xs[r] = (dyn + w[r]) * cmask[r];
            }
        }
        __syncthreads();
    }
    d4 qx = MFMA(qA[0], xs[0], zero4);
    qx = MFMA(qA[1], xs[1], qx);
    qx = MFMA(qA[2], xs[2], qx);
    double part = (xs[0] * (0.5 * qx[0] + qvb[0]) + xs[1] * (0.5 * qx[1] + qvb[1])) + xs[2] * (0.5 * qx[2] + qvb[2]);
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (live && g == 0) a.traj_cost[tj] = cacc + part + gd.q0f;
}

// x0 | padded controls from the handle's pinned staging area into device memory, read over the host link by the kernel itself: a
// copy-engine upload followed by a kernel costs ~10 us of engine hand-over, a kernel followed by a kernel ~2 us
__global__ __launch_bounds__(256) void pets_stage_kernel(const double *__restrict__ src, double *__restrict__ dst, long count) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = src[i];
}

void launch_pets_stage(const double *src, double *dst, long count, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(pets_stage_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, dst, count);
}

__global__ __launch_bounds__(64) void pets_mean_kernel(PetsArgs a) {
    const long ii = blockIdx.x;
    const int l = threadIdx.x;
    double s = 0.0;
    for (long kk = l; kk < a.K; kk += 64) s += a.traj_cost[ii * a.K + kk];
    s = wave_sum(s);
    if (l == 0) a.cost[ii] = s / (double)a.K;
}

void launch_pets(const PetsArgs &a, hipStream_t s) {
    const long ntraj = a.S * a.K;
    if (ntraj <= 0) return;
    // wave16: 0 four trajectories per wavefront; 1 sixteen (MFMA columns), generators split off into their own wavefronts while the launch is
    // too small to occupy every SIMD with recursion waves; 2 sixteen, one wavefront does everything; 3 sixteen, always split
    const long nw = (ntraj + 15) / 16;
    const bool split = a.wave16 == 3 || (a.wave16 == 1 && nw <= PETS_SPLIT_MAX_WAVES);
    if (a.wave16 == 0) hipLaunchKernelGGL(pets_rollout_kernel, dim3((unsigned)((ntraj + 3) / 4)), dim3(64), 0, s, a);
    else if (split) hipLaunchKernelGGL(pets_rollout16s_kernel, dim3((unsigned)nw), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pets_rollout16_kernel, dim3((unsigned)nw), dim3(64), 0, s, a);
    hipLaunchKernelGGL(pets_mean_kernel, dim3((unsigned)a.S), dim3(64), 0, s, a);
}
#endif  // PART_MISC
