// kernels.h -- argument blocks and launchers of kernels.hip (host side: driver.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

struct SweepArgs {
    StateDev st;
    ProblemDev pb;
    OptsDev op;
    int mode;              // 0 gain sweep on nominal slots; 1 policy evaluation of line-search candidates;
                           // 2 policy evaluation of nominal slots with L = 0, mu = 0 (initialize!);
                           // 3 operator form of policy evaluation (sample 0, L/dl given, mu = mu_op)
                           // 4 speculative gain sweep on line-search candidate 0; 5 speculative gain sweep on nominal slots
    int k_first;           // mode 1: first candidate of each sample this launch evaluates (1 when candidate 0 runs in sweep_dual_kernel)
    const double *dl_in;   // mode 3: dl_array or null
    double mu_op;          // mode 3
    double *op_out;        // modes 0/3 operator forms: [0] = s_1, [1] = status ; else null
    double *dump;          // DUMP instantiations: [N+1][DUMP_STRIDE]
    int fly;               // modes 1 / 7: the candidates' records hold only [c_x | c_u | c] (RolloutArgs.notile): f_x | f_u and the cost
                           // Hessian of a step are formed in the sweep from x_t and the problem tables
    int prune;             // round-based path, E > 1: mode 7 publishes whether candidate 0 is the line search's choice (StateDev.acc0); mode 1
                           // launches of candidates 1 .. E-1 poll it and stop (sweep_kernel<.., PRUNE>)
};

// the time-parallel sweep (psweep.h): P waves over P + 1 horizon segments
#define PSW_MAXP 8
struct PswCuts { int P; int cut[PSW_MAXP + 2]; };     // cut[0] = 0 < cut[1] < ... < cut[P + 1] = N

// the time-parallel closed-loop rollout (kernels.hip: rollprl_body): four waves over four horizon segments [cut[w], cut[w+1])
#define PRL_WAVES 4
struct PrlCuts { int cut[PRL_WAVES + 1]; };

struct RolloutArgs {
    StateDev st;
    ProblemDev pb;
    OptsDev op;
    int mode;              // 0 open loop from (x0, u0) into the nominal slots; 1 closed loop candidates
    const double *x0;      // [12]
    const double *u0;      // [N*4]
    double *dump;          // diagnostic builds only
    int notile;            // mode 1, rollin_stage_kernel: candidate records keep only [c_x | c_u | c] (see SweepArgs.fly)
    int multi;             // mode 1 with notile, E <= 16: all candidates of a sample in one wavefront (rollin_multi_kernel)
};

struct FusedArgs {         // solve_fused_kernel: one persistent wavefront per sample (E = 1)
    SweepArgs sw;          // st / pb / op of the batch (mode is set per phase on the device)
    RolloutArgs ro;
    int max_rounds;        // guard on phases per sample
    int dual;              // pair each policy evaluation with the gain sweep that would follow it (sweep_dual_body)
    int occ2;              // one recursion per pass in <= 256 registers: two samples per SIMD (switch fused_occ2; experiment)
    int mat;               // solve_fused_kernel, LQ family with time-invariant cost: tile records are written by the rollouts and read back by
                           // the sweeps (SURVEY 8d's "tiles materialised per trajectory per step"; switch materialize, bench.py's contract leg)
    // the batch's input and outputs, handled by the sample's own wave (no init / gather launches around the solve):
    const double *theta_in;            // [B]; per-sample state is initialised from it (what init_state_kernel does)
    double *out_value;                 // [B] value (Inf for failures) or null
    int *out_status, *out_iters, *out_ls;   // [B] or null
    double *out_cost; double kl_bound; // [B] cost = value + kl_bound / theta  (cross_entropy_bilevel_optimization.jl:193) or null
    // solve_block_kernel only:
    int *census;                       // [CENSUS_SLOTS] per-CU workgroup tickets of the two-wave geometry, or null: see solve_block_kernel
    int helpers;                       // 1: one workgroup per CU (B <= n_cu): the two spare waves of a padded workgroup help linearising
    int acl;                           // 1: closed-loop rollouts of the split geometry in deviation form (rollacl_body: 3 MFMAs on the recursion's chain;
                                       //    values to rounding, not bit-identical to the other paths); 0: the round-2/3 split rollouts (switch block_acl)
    // initialize!'s rollout (ileqg.jl:225-228) does not depend on theta: every sample of a batch -- and every batch on the same (x_0,
    // u_array) -- runs the same open-loop trajectory.  The driver rolls it out ONCE per rat_set_initial into a slot of its own (rollin_kernel,
    // the code the samples would run themselves) and the tile-free kernels copy what they read of it: x, u, the [c_x | c_u | c] rows and
    // the terminal tile.  Null: every sample rolls out for itself.
    const double *init_x, *init_u, *init_t;
    // solve_block_psw_kernel only: segment cuts of its time-parallel sweeps -- two-wave teams (evaluation / gain sweep side by side) and the
    // four-wave team (a sweep that has the compute unit to itself)
    PswCuts psw2e, psw2g, psw4e, psw4g;
    int psw_last;                      // solve_block_kernel, two-wave geometry: the evaluation that ends the solve by both waves, time-parallel (psw2e)
    // solve_block_psw_kernel, two workgroups per sample (see the kernel): duo_stride = B rounded up to a multiple of 8 (0: one workgroup per
    // sample), xw = [Bmax][XW_STRIDE] hand-over words, xepoch = this launch's number on the handle (> 0), duo_count = samples that ran as a pair
    int prl;                           // solve_block_psw_kernel: closed-loop rollouts time-parallel over the workgroup's four waves (rollprl_body;
    PrlCuts prl_cut;                   //  kappa == 0, time-invariant cost; switch psw_prl) and their segment cuts
    int duo_stride;
    unsigned xepoch;
    long long *xw;
    int *duo_count;
};
#define XW_STRIDE 24                   /* 64-bit words per sample: [0] pair word, [8] role A's posts, [16] role B's posts (64 B apart) */
#define CENSUS_SLOTS 4096              /* (XCC_ID, SE_ID, SH_ID, CU_ID) of HW_REG_HW_ID / HW_REG_XCC_ID: 4 + 3 + 1 + 4 bits */

struct LinArgs {
    StateDev st;
    ProblemDev pb;
    int mode;              // 0 nominal slots, 1 candidate slots
};

struct GenDev {                 // device tables of the generative family (PETS)
    int n, m, N, cost_tv, noise_kind;
    const double *Zt, *Ctab, *lin, *q0, *Qf, *qvf;
    double q0f, kappa, l1u, nlo, nhi, tw2;
    const double *nmean, *nchol, *tmean2, *tchol2;   // [16], [12][16] row-major lower
};
struct PetsArgs {
    GenDev g;
    const double *x0;           // [12]
    const double *controls;     // [S][N][4] padded
    long S, K;
    int use_true;
    const double *zn, *zu;      // injected draws or null (device Philox)
    unsigned long long seed;
    long traj0;                 // global index of this launch's first trajectory (device generator counter)
    double *traj_cost;          // [S*K]
    double *cost;               // [S]
    int wave16;                 // 0: pets_rollout_kernel (4 trajectories per wavefront); 1 (default): 16 per wavefront as MFMA columns, noise
                                // generation in separate wavefronts for launches of <= PETS_SPLIT_MAX_WAVES; 2: never split; 3: always split
};
#define PETS_SPLIT_MAX_WAVES 1536
void launch_pets(const PetsArgs &a, hipStream_t s);
void launch_pets_stage(const double *src, double *dst, long count, hipStream_t s);

struct NoisyArgs {              // Monte-Carlo rollouts under process noise (simulate_dynamics with rng)
    ProblemDev pb;
    const double *Wchol;        // [Nw][12][16] lower Cholesky factors of W(k), row-major, zero padded
    const double *xnom;         // [(N+1)][12] nominal states (open loop: only row 0 is used)
    const double *l;            // [N][4]
    const double *L;            // [N][4][12] or null (open loop)
    long K;
    const double *z;            // [K][N][n] injected N(0,1) draws or null (device Philox)
    unsigned long long seed;
    double *x_out, *u_out, *cost;   // [K][(N+1)][12], [K][N][4], [K]; any may be null
    int *dom;                       // [K] DomainError flags or null
};
void launch_noisy_rollout(const NoisyArgs &a, hipStream_t s);

void launch_sweep(const SweepArgs &a, int ntraj, bool gain, bool dump, hipStream_t s);
// the segment-parallel sweep (psweep.h): one workgroup of pc.P wavefronts per trajectory
bool psweep_supported(const SweepArgs &a, bool gain);
void launch_psweep(const SweepArgs &a, int ntraj, bool gain, const PswCuts &pc, hipStream_t s);
void launch_rollout(const RolloutArgs &a, hipStream_t s);
void launch_rollin(const RolloutArgs &a, hipStream_t s);      // fused rollout + linearise (solver hot loop)
void launch_linearize(const LinArgs &a, hipStream_t s);
void launch_solve_fused(const FusedArgs &a, hipStream_t s);   // complete solve! per sample in one launch (E = 1)
void launch_solve_block(const FusedArgs &a, hipStream_t s);   // complete solve! per sample by a workgroup of wavefronts (E = 1, 2, 4, 8)
bool solve_block_psw_supported(const FusedArgs &a);
void launch_solve_block_psw(const FusedArgs &a, hipStream_t s);   // ... with time-parallel sweeps: four wavefronts per sample, one sample per CU (E = 1)
bool solve_block_supported(int E);
void launch_init_state(const StateDev &st, const OptsDev &op, const double *theta_dev, hipStream_t s);
void launch_copy_initial(const StateDev &st, const double *init_x, const double *init_u, const double *init_t, hipStream_t s);   // round-based path: FusedArgs.init_* per sample
void launch_ls_select(const StateDev &st, const OptsDev &op, int slot, hipStream_t s);
void launch_sweep_cand0(const SweepArgs &a, int nsamples, hipStream_t s);  // mode 7 on tile-free candidates: the paired pass, or the plain evaluation where accepting would end solve!
void launch_sweep_dual(const SweepArgs &a, int nsamples, hipStream_t s);   // modes 6 (initialize! + first gain sweep), 7 (candidate 0 + next gain sweep)
void launch_commit_init(const StateDev &st, hipStream_t s);
bool rollin_notile_supported(const ProblemDev &pb, const StateDev &st);              // the speculative path can run without candidate tiles
void launch_pack_solution(const StateDev &st, int b, double *dst_x, double *dst_u, double *dst_L, double *dst_s, hipStream_t s);   // one solve's outputs, one launch
void launch_gather(const StateDev &st, double *value, int *status, int *iters, int *ls_evals, double *cost, double kl_bound, hipStream_t s);
