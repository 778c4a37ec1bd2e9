// layout.h -- HBM data layout shared by the HIP kernels and the host driver.
//
// The kernels are specialised for the padded size NP = 12 states, MP = 4 controls (n <= 12, m <= 4):
// NP + MP = 16 is exactly the edge of one v_mfma_f64_16x16x4_f64 tile, so [A|B] is a 12x16 operand and
// the Q-function Hessian [[Q,P'],[P,R]] is one 16x16 accumulator.  Smaller problems are embedded with
// zero padding (identity on the padded diagonals of W^-1 and R, which leaves every result unchanged).
//
// Tile bundle of one trajectory (what approximate_model, ileqg.jl:258-322, produces).  One step record is a REGISTER IMAGE
// of what the backward sweep keeps per lane, so that the sweep fetches it with three 16-B/lane loads and two 8-B/lane
// loads (a VMEM instruction costs a single-wave-per-SIMD kernel ~40 issue cycles whatever its width: fewer, wider):
//   logical registers (64 doubles each, element 64 r + l = row 4 r + (l >> 4), column l & 15 of a 16-wide row-major block)
//     R0..R2  Z = [A | B]              12 x 16   (f_x | f_u)
//     R3..R6  C = [[Q, *], [P, R]]     16 x 16   (c_xx | unused ; c_ux | c_uu); the 48 slots of rows 0..11, columns 12..15
//                                                are never consumed by the sweep (written as zeros)
//   physical record, step t = 0..N-1, stride TSTRIDE = 420 doubles:
//     [  0 .. 128)  registers R0, R1 interleaved per lane: position 2 l + h                (one global_load_dwordx4)
//     [128 .. 256)  registers R2, R6 interleaved per lane: position 128 + 2 l + h          (one global_load_dwordx4)
//     [256 .. 352)  registers R3, R4 (rows 0..7 of C), live lanes only, interleaved: position 256 + 2 (12 g + j) + h for
//                   lane (g, j), j < 12                                                     (one global_load_dwordx4)
//     [352 .. 400)  register R5 (rows 8..11 of C), live lanes only: position 352 + 12 g + j  (one global_load_dwordx2)
//     [400 .. 416)  qr = [q_vec | r]  (c_x | c_u)          [416] q = c          [417] unused
//     [418 .. 420)  two zeros: what the dead lanes (columns 12..15 of R3..R5) load and store, so that every access stays
//                   unconditional and full-width (16-B aligned)
//   terminal block at N*TSTRIDE: Qf 12x12 row-major (144), q_vec (12), q (1), pad (1)  -> TTERM = 158
// The information content is SURVEY.md section 8's n^2 + nm + n^2 + m^2 + mn + n + m + 1 = 417 doubles per step
// (+157 terminal = 168,056 B at N = 50); the physical record carries 3 more (+0.7 %).
#pragma once

#define RAT_NP 12
#define RAT_MP 4
#define RAT_PD 16
#define RAT_AUG 12          /* index of the homogeneous coordinate in the augmented value matrix */

#define TS_C34 256
#define TS_R5  352
#define TS_QR  400
#define TS_q   416
#define TS_PAD 418          /* two zeros (16-B aligned): target of the dead lanes and of idle-lane stores */
#define TSTRIDE 420
#define TS_INFO 417         /* doubles of information per step */
#define TT_Q   0
#define TT_QV  144
#define TT_q   156
#define TTERM  158
#define TT_INFO 157

/* physical position of lane l of logical register R (0..6); dead lanes of R3..R5 map to the zero pair */
#define TS_LIVE(l) (((l) & 15) < 12)
#define TS_CL(l) (((l) >> 4) * 12 + ((l) & 15))
#define TS_REG(R, l) ((R) < 2 ? 2 * (l) + (R) : (R) == 2 ? 128 + 2 * (l) : (R) == 6 ? 129 + 2 * (l) : \
                      (R) < 5 ? (TS_LIVE(l) ? TS_C34 + 2 * TS_CL(l) + ((R) - 3) : TS_PAD + ((R) - 3)) : \
                                (TS_LIVE(l) ? TS_R5 + TS_CL(l) : TS_PAD))
/* [A|B] row i (0..11), column c (0..15);  C row i (0..15), column c (0..15) */
#define TS_ZPOS(i, c) TS_REG(((i) * 16 + (c)) >> 6, ((i) * 16 + (c)) & 63)
#define TS_CPOS(i, c) TS_REG(3 + (((i) * 16 + (c)) >> 6), ((i) * 16 + (c)) & 63)

#define XSTR 12             /* doubles per time step of a state history   */
#define USTR 4              /* doubles per time step of a control history */
#define LSTR 48             /* doubles per time step of a gain history: L_t as 4 x 12 row-major */

#define DUMP_S   0          /* per-step debug dump (rat_dp_* operators): S 12x12 row-major */
#define DUMP_SV  144
#define DUMP_s   156
#define DUMP_g   157
#define DUMP_G   161        /* 4 x 12 row-major */
#define DUMP_H   209        /* 4 x 4 row-major  */
#define DUMP_STRIDE 225

/* LDS scratch of the phase bodies, in doubles.  The bodies take pointers (no static __shared__ of their own): a kernel that inlines
 * several of them -- or runs them on several wavefronts of one workgroup -- declares ONE area per wave instead of one per instantiation. */
#define WLS_SWEEP 232       /* sweep_body: gain rows [L|dl] 64 + exchange area 168 */
#define WLS_DUAL  400       /* sweep_dual_body: 64 + 2 x 168 */
#define ROLLIN_NST 52       /* longest horizon whose closed-loop operands are staged in LDS by the fused solves */
#define STG_CL ((ROLLIN_NST * LSTR + 63) / 64)
#define STG_CX (((ROLLIN_NST + 1) * XSTR + 63) / 64)
#define STG_CU ((ROLLIN_NST * USTR + 63) / 64)
#define STG_DOUBLES ((STG_CL + STG_CX + 2 * STG_CU) * 64)   /* L, xbar, l, dl of one trajectory: 29,184 B */
#define STG_PAD (STG_CL * 64)  /* doubles of slack behind the x / u / L / dl pools: whole-chunk staging loads may run past the last slot */

#define ST_RUNNING (-1)
#define CTR_RING 8          /* per-round counter pairs kept in a ring (host polls one round behind) */

struct ProblemDev {
    int model, n, m, N, cost_tv, W_tv;
    int W_diag;          // 1: W is time-invariant and diagonal (the sweeps then fold inv(W) into M^-1: see sweep_body)
    const double *Zt;    // [192]      [A|B] 12x16 row-major, zero padded
    const double *Ctab;  // [Nc][256]  [[Q,P'],[P,R]] 16x16 row-major, padded R diagonal = 1
    const double *lin;   // [Nc][16]   [qv | rv]
    const double *q0;    // [Nc]
    const double *Qf;    // [144]      12x12 row-major
    const double *qvf;   // [16]
    double q0f, kappa;
    double pl_a, pl_b, pl_p, pl_pu, pl_cx, pl_cu, pl_h;
    const double *Winv;  // [Nw][192]  inv(W(k)) 12x16 row-major, padded diagonal = 1
    const double *Wp;    // [Nw][192]  W(k), zero padded
    const double *epiv;  // [Nw][16]   even k: 1/(e_k e_k+1), e = elimination pivots of the padded inv(W(k)); odd k: 1 (logdet pairing)
    const double *logdetW; // [Nw]
    const double *Wdg;   // [16]       W_diag: the diagonal of inv(W), padded entries 1
};

// All per-sample / per-slot device state of one handle.
struct StateDev {
    int B, E, N;
    int tile_alias;                            // 1: one tile bundle per SAMPLE instead of per slot (single-launch E = 1 path).  The
                                               // nominal trajectory's tiles are dead once the gain sweep of its iteration has run
                                               // (line_search! never reads them, ileqg.jl:494-592; step! re-linearises what was
                                               // accepted, :604), so every candidate is linearised over them in place: the tile
                                               // traffic of a whole batch (173 MB at B = 1024) stays inside the 256 MB MALL
    long tile_stride, x_stride, u_stride;      // doubles per slot
    double *tiles, *xs, *us;                   // slot pools: B*(E+1) slots (tiles: B bundles when tile_alias)
    double *L, *dl;                            // [2][Bmax][N*48], [2][Bmax][N*4]: double-buffered gains (lsel[b] = live half)
    long l_half, dl_half;                      // doubles per half
    int *lsel;                                 // [B] which half holds the committed L_array / dl of sample b
    double *mu_spec, *delta_spec;              // [B] mu, Delta left by the speculative gain sweep
    int *spec_st;                              // [B] 0 none, 1 speculative gain sweep valid, 2 it hit M not PD, 5 mu diverged
    double *theta, *mu, *delta, *value, *d_cur, *eps_init, *ls_eps;
    int *status, *iter, *ls_active, *ls_count, *slot_nom, *n_ls, *hist_n;
    double *value_c, *d_c;                     // [B*E]
    int *flag_c;                               // [B*E] 0 ok, 1 DP failed (M not PD), 2 domain failure
    int *acc0;                                 // [B] round-based path, E > 1: 1 once candidate 0 of this round is known to be the line search's choice
                                               // (the paired sweep that evaluates it says so): the evaluations of candidates 1 .. E-1 of the
                                               // sample, whose results the sequential rule would then never read, stop where they are
    double *hist; int hist_cap;                // [B][2*hist_cap] or null
    int *counters;                             // [CTR_RING][2]: per round {samples still in line search, samples running}
    double *sink;                              // [SINK_SLOTS][64] write-only: idle lanes of unconditional stores (a lane-conditional store
                                               // splits the basic block the scheduler works on); one slot per sample (mod SINK_SLOTS): a
                                               // single slot would have every wavefront of a launch write the same cache lines on every step
};
#define SINK_SLOTS 1024
__host__ __device__ inline double *sample_sink(const StateDev &st, int b) { return st.sink + (long)(b & (SINK_SLOTS - 1)) * 64; }

struct OptsDev {
    double mu_min, delta_0, lambda, d, eps_init, eps_min;
    int iter_max, adaptive;
};

__host__ __device__ inline long tile_slot(const StateDev &st, int b, int slot) { return st.tile_alias ? (long)b : (long)slot; }
__host__ __device__ inline int cand_slot(int b, int k, int nom, int E) {
    int s = (k < nom) ? k : k + 1;
    return b * (E + 1) + s;
}
