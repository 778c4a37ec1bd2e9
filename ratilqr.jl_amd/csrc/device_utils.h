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
#define DIAG_DECL unsigned long long dg_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long dg_prev = 0, dg_gap = 0;
#define DIAG_START() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        { unsigned long long now_ = __builtin_readcyclecounter(); if (dg_prev) dg_gap += now_ - dg_prev; dg_prev = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define DIAG_STAMP(i, val) do { asm volatile("" :: "v"(val)); __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        { unsigned long long now_ = __builtin_readcyclecounter(); dg_acc[i] += now_ - dg_prev; dg_prev = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIAG_DECL
#define DIAG_START() do {} while (0)
#define DIAG_STAMP(i, val) do {} while (0)
#endif

