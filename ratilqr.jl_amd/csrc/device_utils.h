// device_utils.h -- small device helpers shared by the HIP translation units (kernels.hip, sweep_dual.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ d4 mm3(const d4 &a, const d4 &b, d4 acc) {     // acc += A' B over K-slices 0..2 (rows 0..11)
    acc = MFMA(a[0], b[0], acc);
    acc = MFMA(a[1], b[1], acc);
    acc = MFMA(a[2], b[2], acc);
    return acc;
}

// Single-wavefront workgroups: LDS operations of one wave complete in issue order, so a cross-lane exchange
// through LDS needs no s_barrier and -- unlike __syncthreads() -- must not drain vmcnt (that would serialise the
// software-prefetched tile loads of the next time step).  This only stops the compiler from reordering.
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// 1/p: v_rcp_f64 seed + two Newton steps (relative error ~1e-16; the full IEEE divide sequence is ~2x longer and sits
// on the serial pivot chain of the elimination)
__device__ __forceinline__ double fast_rcp(double p) {
    double x = __builtin_amdgcn_rcp(p);
    double e = fma(-p, x, 1.0);
    x = fma(x, e, x);
    e = fma(-p, x, 1.0);
    x = fma(x, e, x);
    return x;
}

// pivot-chain variant: one Newton step (measured max relative error 2.1e-15, tools/ubench/mfma_rcp.hip)
__device__ __forceinline__ double fast_rcp1(double p) {
    double x = __builtin_amdgcn_rcp(p);
    const double e = fma(-p, x, 1.0);
    return fma(x, e, x);
}

// Loads of data ANOTHER COMPUTE UNIT may have written during this kernel (the two-workgroups-per-sample solve, kernels.hip: a partner
// workgroup in the same XCD hands trajectories, gains and control words over through the XCD's L2).  A compute unit's vector L1 is
// write-through but is not invalidated by another unit's stores, and the agent-scope acquire that would invalidate it (buffer_inv sc1) walks
// the L2: 3.6 us each (tools/ubench/xwg_handoff.hip).  An agent-scope relaxed ATOMIC load carries sc1 and misses the L1 by definition --
// a 13 KB hand-over costs ~1 us this way.  Only the translation unit of that kernel defines RAT_XC; everywhere else xld is a plain load.
template <class T>
__device__ __forceinline__ T xld(const T *p) {
#ifdef RAT_XC
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}

// A per-sample index every lane loaded from the same address: telling the compiler it is wave-uniform (v_readfirstlane) moves the
// slot / pointer arithmetic built on it, and the per-step address updates of the time loops, from the vector ALU to the scalar unit.
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// sum over each 16-lane row with DPP row rotations (VALU speed; no trip through the LDS crossbar like ds_bpermute)
__device__ __forceinline__ double row_sum16(double x) {
#define ROW_ROR(v, n) __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + (n), 0xF, 0xF, false), \
                                       __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + (n), 0xF, 0xF, false))
    x += ROW_ROR(x, 8);
    x += ROW_ROR(x, 4);
    x += ROW_ROR(x, 2);
    x += ROW_ROR(x, 1);
#undef ROW_ROR
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One 2x2-block round of the symmetric sweep (in-place inversion of M = inv(W) - theta S, ileqg.jl:365-367) with the rank-2
// update on the matrix pipe.  m holds M (12 x 12 in the 12 x 16 accumulator layout, register r = rows 4r..4r+3).
//   P = M_KK (K = {k, k+1}, k = 2 KB), Bk = P^-1:   M'_KK = -Bk,  M'_Kj = Bk M_Kj,  M'_iK = M_iK Bk,  M'_ij = M_ij - M_iK Bk M_Kj
// With the pivot rows t (other rows zeroed, -I written into the pivot block) every entry obeys the ONE formula
//   M' = M o mask - t' (Bk t)            t: 2 x 16 pivot rows,  t' its transpose (M is symmetric: M_iK = (M_Ki)')
// which is a K = 2 contraction: v_mfma_f64_16x16x4 with A = t read as the A operand (lane (g, j) of the register that holds the
// pivot rows IS A[row j][k-slot g]: no data movement), B = -(Bk t) computed on the pivot-row lanes (each needs its partner row's
// entry: one v_permlane16_swap exchanges rows 0<->1, 2<->3), C = M o mask.  No LDS, no fence, ~30 instructions per round.
// Leading minors p11 > 0, det P > 0 for every block  <=>  isposdef(M)  (:366); det P = d_k d_{k+1} feeds logdet.
// ---------------------------------------------------------------------------------------------------------------------------
struct ElimMasks {
    double tm[6], wa[6];        // pivot rows of round kb: tm = 1 on (pivot row, non-pivot column), wa = -1 on the pivot block's diagonal
    double cm[6], crm[6];       // cm = 0 on the pivot columns; crm = 0 on pivot columns and (register of the pivot rows) pivot rows
    double e0[2], e1[2];        // row selectors for kg = 0 / 2: e0 = 1 on row kg, e1 = 1 on row kg + 1
    bool odd;                   // lane sits in an odd 16-lane row
};
__device__ __forceinline__ void elim_masks(ElimMasks &em, int g, int j) {
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) {
        const int k = 2 * kb, kg = k & 3;
        const bool colk = (j == k) || (j == k + 1), rowk = (g == kg) || (g == kg + 1);
        em.tm[kb] = (rowk && !colk) ? 1.0 : 0.0;
        em.wa[kb] = (rowk && j == k + (g - kg)) ? -1.0 : 0.0;
        em.cm[kb] = colk ? 0.0 : 1.0;
        em.crm[kb] = (colk || rowk) ? 0.0 : 1.0;
    }
    em.e0[0] = (g == 0) ? 1.0 : 0.0; em.e1[0] = (g == 1) ? 1.0 : 0.0;
    em.e0[1] = (g == 2) ? 1.0 : 0.0; em.e1[1] = (g == 3) ? 1.0 : 0.0;
    em.odd = (g & 1) != 0;
}
// The same column of the adjacent row (0<->1, 2<->3), two ways with identical results:
//   SWZ = true : ds_swizzle bit mode, lane ^= 16 inside each half of the wave (and 0x1f, or 0, xor 0x10) -- the LDS crossbar without
//                memory: one instruction per word and the result IS the partner, but an LDS round trip of latency.  For the paired
//                recursions, whose two chains hide each other's latency and which are short of issue slots.
//   SWZ = false: v_permlane16_swap (gfx950) of the register with itself, then a select by row parity: two copies, a swap and a
//                select per word, all VALU (short latency).  For the single recursion, which is bound by its dependency chain.
// XCH selects how a round moves data across lanes (identical values every way):
//   0  v_permlane16_swap row exchange, pivot block through v_readlane       -- all vector ALU, shortest latency: single recursions
//   1  ds_swizzle row exchange (LDS crossbar), pivot block through v_readlane -- fewer vector instructions: paired recursions, OCC2
// (Tried and dropped, round 3: the pivot block through ds_bpermute instead of v_readlane -- it lands in vector registers, so the round
//  loses its six v_readlane and the v_mov the constant-bus limit forces: 15 instead of 23 vector instructions.  The saturated E = 8
//  evaluation launch did not move: 1.301 -> 1.294 ms per batch, 232 instead of 202 registers.)
template <int XCH>
__device__ __forceinline__ double row_partner(double x, bool odd) {
    if (XCH)
        return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F));
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // [0] = (r0, r0, r2, r2), [1] = (r1, r1, r3, r3)
    const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)(odd ? sh[0] : sh[1]), (int)(odd ? sl[0] : sl[1]));
}
// Leading minors p11 > 0 and det P > 0 of every block <=> isposdef(M) (:366).  A non-NaN double is > 0 iff its high word, read as a
// signed integer, is > 0 (a positive subnormal below 2^-1022 * 2^-20 counts as singular), so the running minimum is an integer
// minimum over high words: one SALU op for p11 (it lives in SGPRs) and one VALU op for det -- v_min_f64 would cost three with the
// canonicalisation that fmin() carries.  NaNs (and infinities) are caught through the running product of the determinants, rprod, which
// the caller tests once per step.
template <int KB, int XCH>
__device__ __forceinline__ void elim_round(d4 &m, const ElimMasks &em, int &pdmin, double &rprod) {
    constexpr int k = 2 * KB, kr = k >> 2, kg = k & 3;          // rows k, k+1 live in register kr, quad-rows kg, kg+1
    const double p11 = readlane_f64(m[kr], kg * 16 + k);
    const double p12 = readlane_f64(m[kr], kg * 16 + k + 1);
    const double p22 = readlane_f64(m[kr], (kg + 1) * 16 + k + 1);
    const double t = fma(m[kr], em.tm[KB], em.wa[KB]);          // pivot rows, -I in the pivot block, zero elsewhere
    const double other = row_partner<XCH>(t, em.odd);
    const double det = fma(p11, p22, -(p12 * p12));
    const double idet = fast_rcp1(det);
    pdmin = min(pdmin, min(__double2hiint(p11), __double2hiint(det)));
    rprod *= det;            // logdet(W M) = sum_k log(det P_k / (e_k e_k+1))  (:387): wave-uniform running product; the caller
                             // multiplies the step's prod_k 1/(e_k e_k+1) in once (before the rounds) and renormalises per step
    // -(Bk t) on the pivot-row lanes (Bk = adj(P) / det): row k: -(p22 t_k - p12 t_k+1) / det, row k+1: -(p11 t_k+1 - p12 t_k) / det
    const double pd = em.e0[kg >> 1] * p22 + em.e1[kg >> 1] * p11;
    const double nu = fma(p12, other, -(pd * t)) * idet;
    m[0] *= (kr == 0 ? em.crm[KB] : em.cm[KB]);                 // C operand in place (rows 12..15, register 3, stay zero)
    m[1] *= (kr == 1 ? em.crm[KB] : em.cm[KB]);
    m[2] *= (kr == 2 ? em.crm[KB] : em.cm[KB]);
    m = MFMA(t, nu, m);
}

// f_x of the LQ family, one register of the [A | B] image: A_ij, + 3 kappa x_i^2 on the lane that holds a diagonal element (dg = 1 there,
// 0 elsewhere; x = that lane's own state component).  Formed by the rollout kernels (tile records) and the fly sweeps (in registers):
// one rounding order everywhere, whatever the surrounding code lets the compiler contract.
__device__ __forceinline__ double fx_diag(double zt, double dg, double kappa, double x) {
#pragma clang fp contract(off)
    const double k3 = 3.0 * kappa;
    const double xx = x * x;
    const double d = k3 * xx;
    return __builtin_fma(dg, d, zt);
}

// One term of c = [x;u]' (1/2 C [x;u] + lin) + q0 (ileqg.jl:296): xu (1/2 cxu + lin), one rounding order in every rollout kernel
__device__ __forceinline__ double cost_term(double xu, double cxu, double lin) {
#pragma clang fp contract(off)
    const double h = __builtin_fma(0.5, cxu, lin);
    return xu * h;
}

// The four 16-lane rows of x, each broadcast to every row (v_permlane32_swap + v_permlane16_swap, gfx950; tools/ubench/xlane.hip checks
// the pattern lane by lane): r[g] on lane (., j) = x of lane (g, j).
__device__ __forceinline__ void rows_bcast(double x, double (&r)[4]) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto s32l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);        // [0] = rows (0, 1, 0, 1), [1] = rows (2, 3, 2, 3)
    const auto s32h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const auto al = __builtin_amdgcn_permlane16_swap(s32l[0], s32l[0], false, false);   // [0] = row 0 everywhere, [1] = row 1 everywhere
    const auto ah = __builtin_amdgcn_permlane16_swap(s32h[0], s32h[0], false, false);
    const auto bl = __builtin_amdgcn_permlane16_swap(s32l[1], s32l[1], false, false);   // rows 2, 3
    const auto bh = __builtin_amdgcn_permlane16_swap(s32h[1], s32h[1], false, false);
    r[0] = __hiloint2double((int)ah[0], (int)al[0]); r[1] = __hiloint2double((int)ah[1], (int)al[1]);
    r[2] = __hiloint2double((int)bh[0], (int)bl[0]); r[3] = __hiloint2double((int)bh[1], (int)bl[1]);
}

// One round of TWO independent eliminations (the paired recursions of sweep_dual_body), phased so that neither chain waits for its row
// exchange: both pull their pivot block and pivot rows and issue the ds_swizzle first, run the wave-uniform pivot arithmetic of both
// (determinant, reciprocal, the partner-free half of nu) under the crossbar's latency, and only then touch the exchanged rows.  The
// operations and their operands are those of elim_round<KB, 1> for each chain -- identical values; only their order in the wave's
// instruction stream differs.  (Issued chain after chain, every round parked the wave on lgkmcnt(0) five instructions after its swizzle.)
template <int KB>
__device__ __forceinline__ void elim_round_pair(d4 &mA, d4 &mB, const ElimMasks &em, int &pdminA, int &pdminB, double &rprodA, double &rprodB) {
    constexpr int k = 2 * KB, kr = k >> 2, kg = k & 3;
    const double a11 = readlane_f64(mA[kr], kg * 16 + k), a12 = readlane_f64(mA[kr], kg * 16 + k + 1), a22 = readlane_f64(mA[kr], (kg + 1) * 16 + k + 1);
    const double tA = fma(mA[kr], em.tm[KB], em.wa[KB]);
    const double otherA = row_partner<1>(tA, em.odd);
    const double b11 = readlane_f64(mB[kr], kg * 16 + k), b12 = readlane_f64(mB[kr], kg * 16 + k + 1), b22 = readlane_f64(mB[kr], (kg + 1) * 16 + k + 1);
    const double tB = fma(mB[kr], em.tm[KB], em.wa[KB]);
    const double otherB = row_partner<1>(tB, em.odd);
    __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise regroups the round chain by chain)
    const double detA = fma(a11, a22, -(a12 * a12));
    const double idetA = fast_rcp1(detA);
    pdminA = min(pdminA, min(__double2hiint(a11), __double2hiint(detA)));
    rprodA *= detA;
    const double pdA = em.e0[kg >> 1] * a22 + em.e1[kg >> 1] * a11;
    const double ptA = pdA * tA;
    const double detB = fma(b11, b22, -(b12 * b12));
    const double idetB = fast_rcp1(detB);
    pdminB = min(pdminB, min(__double2hiint(b11), __double2hiint(detB)));
    rprodB *= detB;
    const double pdB = em.e0[kg >> 1] * b22 + em.e1[kg >> 1] * b11;
    const double ptB = pdB * tB;
    __builtin_amdgcn_sched_barrier(0);
    mA[0] *= (kr == 0 ? em.crm[KB] : em.cm[KB]);
    mA[1] *= (kr == 1 ? em.crm[KB] : em.cm[KB]);
    mA[2] *= (kr == 2 ? em.crm[KB] : em.cm[KB]);
    const double nuA = fma(a12, otherA, -ptA) * idetA;
    mA = MFMA(tA, nuA, mA);
    mB[0] *= (kr == 0 ? em.crm[KB] : em.cm[KB]);
    mB[1] *= (kr == 1 ? em.crm[KB] : em.cm[KB]);
    mB[2] *= (kr == 2 ? em.crm[KB] : em.cm[KB]);
    const double nuB = fma(b12, otherB, -ptB) * idetB;
    mB = MFMA(tB, nuB, mB);
}

// The scalars of a sweep that never feed back, folded once at its end: 0.5 sum(racc) - logdet(W M) / (2 theta) with the running product of the
// block determinants renormalised to mantissa x 2^rexp.  ONE rounding order in every kernel whose results must agree bit for bit (the
// compiler's own choice between a fused and a separate multiply-add depends on the code around the expression: it changed in one of the two
// compilations of the paired sweep when an unrelated statement was added behind it, on the samples whose exponent sum is not zero).
__device__ __forceinline__ double sweep_scalars(double racc_sum, double coef, double rprod, int rexp, bool risk) {
#pragma clang fp contract(off)
    const double half = 0.5 * racc_sum;
    if (!risk) return half;
    const double ld = __builtin_fma((double)rexp, 0.6931471805599453094, log(rprod));
    const double t = coef * ld;
    return half + t;
}

__device__ __forceinline__ bool isapprox_default(double x, double y) {   // Base.isapprox, rtol = sqrt(eps), atol = 0
    if (x == y) return true;
    if (!isfinite(x) || !isfinite(y)) return false;
    return fabs(x - y) <= 1.4901161193847656e-8 * fmax(fabs(x), fabs(y));
}

// Diagnostic build only (make diag): s_memtime stamps per segment of the time step; shares of one wave's cycles are
// written to the dump buffer.  Never compiled into the product library.
// Phase-timeline build only (make diagp): absolute cycle stamps at the entry / loop start / loop end / exit of the phase bodies
#ifdef RAT_DIAG_PHASES
#define BODY_MARK(dump_, slot_) do { if (threadIdx.x == 0 && blockIdx.x < 8 && (dump_)) \
        (dump_)[640 + blockIdx.x * 32 + (slot_)] = (double)__builtin_readcyclecounter(); } while (0)
#else
#define BODY_MARK(dump_, slot_) do {} while (0)
#endif

#ifdef RAT_DIAG
#define DIAG_DECL unsigned long long dg_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long dg_prev = 0; [[maybe_unused]] unsigned long long dg_gap = 0;
#define DIAG_START() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        { unsigned long long now_ = __builtin_readcyclecounter(); if (dg_prev) dg_gap += now_ - dg_prev; dg_prev = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define DIAG_STAMP(i, val) do { asm volatile("" :: "v"(val)); __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        { unsigned long long now_ = __builtin_readcyclecounter(); dg_acc[i] += now_ - dg_prev; dg_prev = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIAG_DECL
#define DIAG_START() do {} while (0)
#define DIAG_STAMP(i, val) do {} while (0)
#endif

