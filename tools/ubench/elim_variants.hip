// The elimination round of the sweeps (M = inv(W) - theta S inverted in place, ileqg.jl:365-367; device_utils.h: elim_round), measured
// against the variants VERDICT r05 next #5 names -- one wavefront per SIMD, every elimination's input depending on the previous one's
// output (the sweep's situation: latency-bound), and two wavefronts per SIMD (issue-bound: the two-samples-per-SIMD kernel).
//   V0  today's round: 2 x 2 block pivots, 6 rounds, pivot block through v_readlane x6, row exchange by v_permlane16_swap
//   V1  1 x 1 pivots, 12 rounds (13 vector instructions + 1 MFMA each)
//   V2  1 x 1 pivots with the NEXT pivot taken from the Schur complement's diagonal before the round's MFMA lands
//       (p' = m[k+1][k+1] - m[k+1][k]^2 / p: its reciprocal chain runs under the MFMA's latency)
//   V3  today's round with the bookkeeping (determinant product for logdet, leading-minor minimum) folded to every second round
// Reported: cycles per elimination (s_memtime, shader clock), vector-instruction counts are taken from the disassembly
// (llvm-objdump -d, the loop bodies between the s_memtime pairs).  Results of V1 / V2 are checked against V0's inverse.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../ratilqr.jl_amd/csrc elim_variants.hip -o elim_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "device_utils.h"

struct Masks1 { double tm[12], wa[12], cm[12], crm[12]; };
__device__ __forceinline__ void masks1(Masks1 &mk, int g, int j) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int kg = k & 3;
        mk.tm[k] = (g == kg && j != k) ? 1.0 : 0.0;
        mk.wa[k] = (g == kg && j == k) ? -1.0 : 0.0;
        mk.cm[k] = (j == k) ? 0.0 : 1.0;
        mk.crm[k] = (j == k || g == kg) ? 0.0 : 1.0;
    }
}
// one 1 x 1 round: M' = M o mask - t' (t / p), t = pivot row with -1 in the pivot position
template <int K>
__device__ __forceinline__ void round1(d4 &m, const Masks1 &mk, int &pdmin, double &rprod) {
    constexpr int kr = K >> 2, kg = K & 3;
    const double p = readlane_f64(m[kr], kg * 16 + K);
    const double t = fma(m[kr], mk.tm[K], mk.wa[K]);
    const double ip = fast_rcp1(p);
    pdmin = min(pdmin, __double2hiint(p));
    rprod *= p;
    const double nu = -(t * ip);
    m[0] *= (kr == 0 ? mk.crm[K] : mk.cm[K]);
    m[1] *= (kr == 1 ? mk.crm[K] : mk.cm[K]);
    m[2] *= (kr == 2 ? mk.crm[K] : mk.cm[K]);
    m = MFMA(t, nu, m);
}
// ... with the next round's pivot formed ahead: pn = m[k+1][k+1] - m[k+1][k]^2 / p (read before the MFMA), handed on
template <int K>
__device__ __forceinline__ void round1_ahead(d4 &m, const Masks1 &mk, int &pdmin, double &rprod, double &p) {
    constexpr int kr = K >> 2, kg = K & 3;
    constexpr int K1 = K + 1, kr1 = K1 >> 2, kg1 = K1 & 3;
    const double t = fma(m[kr], mk.tm[K], mk.wa[K]);
    const double ip = fast_rcp1(p);
    pdmin = min(pdmin, __double2hiint(p));
    rprod *= p;
    double pn = 1.0;
    if (K < 11) {
        const double a = readlane_f64(m[kr1], kg1 * 16 + K1), c = readlane_f64(m[kr1], kg1 * 16 + K);
        pn = fma(-(c * c), ip, a);
    }
    const double nu = -(t * ip);
    m[0] *= (kr == 0 ? mk.crm[K] : mk.cm[K]);
    m[1] *= (kr == 1 ? mk.crm[K] : mk.cm[K]);
    m[2] *= (kr == 2 ? mk.crm[K] : mk.cm[K]);
    m = MFMA(t, nu, m);
    p = pn;
}
// today's round without its bookkeeping (V3: the caller folds determinant / minimum of two rounds into one update)
template <int KB>
__device__ __forceinline__ void round2_nobook(d4 &m, const ElimMasks &em, double &p11o, double &deto) {
    constexpr int k = 2 * KB, kr = k >> 2, kg = k & 3;
    const double p11 = readlane_f64(m[kr], kg * 16 + k);
    const double p12 = readlane_f64(m[kr], kg * 16 + k + 1);
    const double p22 = readlane_f64(m[kr], (kg + 1) * 16 + k + 1);
    const double t = fma(m[kr], em.tm[KB], em.wa[KB]);
    const double other = row_partner<0>(t, em.odd);
    const double det = fma(p11, p22, -(p12 * p12));
    const double idet = fast_rcp1(det);
    p11o = p11; deto = det;
    const double pd = em.e0[kg >> 1] * p22 + em.e1[kg >> 1] * p11;
    const double nu = fma(p12, other, -(pd * t)) * idet;
    m[0] *= (kr == 0 ? em.crm[KB] : em.cm[KB]);
    m[1] *= (kr == 1 ? em.crm[KB] : em.cm[KB]);
    m[2] *= (kr == 2 ? em.crm[KB] : em.cm[KB]);
    m = MFMA(t, nu, m);
}

template <int V>
__device__ __forceinline__ void eliminate(d4 &m, const ElimMasks &em, const Masks1 &mk, int &pdmin, double &rprod) {
    if (V == 0) {
        elim_round<0, 0>(m, em, pdmin, rprod); elim_round<1, 0>(m, em, pdmin, rprod); elim_round<2, 0>(m, em, pdmin, rprod);
        elim_round<3, 0>(m, em, pdmin, rprod); elim_round<4, 0>(m, em, pdmin, rprod); elim_round<5, 0>(m, em, pdmin, rprod);
    } else if (V == 1) {
        round1<0>(m, mk, pdmin, rprod); round1<1>(m, mk, pdmin, rprod); round1<2>(m, mk, pdmin, rprod); round1<3>(m, mk, pdmin, rprod);
        round1<4>(m, mk, pdmin, rprod); round1<5>(m, mk, pdmin, rprod); round1<6>(m, mk, pdmin, rprod); round1<7>(m, mk, pdmin, rprod);
        round1<8>(m, mk, pdmin, rprod); round1<9>(m, mk, pdmin, rprod); round1<10>(m, mk, pdmin, rprod); round1<11>(m, mk, pdmin, rprod);
    } else if (V == 2) {
        double p = readlane_f64(m[0], 0);
        round1_ahead<0>(m, mk, pdmin, rprod, p); round1_ahead<1>(m, mk, pdmin, rprod, p); round1_ahead<2>(m, mk, pdmin, rprod, p);
        round1_ahead<3>(m, mk, pdmin, rprod, p); round1_ahead<4>(m, mk, pdmin, rprod, p); round1_ahead<5>(m, mk, pdmin, rprod, p);
        round1_ahead<6>(m, mk, pdmin, rprod, p); round1_ahead<7>(m, mk, pdmin, rprod, p); round1_ahead<8>(m, mk, pdmin, rprod, p);
        round1_ahead<9>(m, mk, pdmin, rprod, p); round1_ahead<10>(m, mk, pdmin, rprod, p); round1_ahead<11>(m, mk, pdmin, rprod, p);
    } else {
        double pa, da, pb, db;
        round2_nobook<0>(m, em, pa, da); round2_nobook<1>(m, em, pb, db);
        pdmin = min(pdmin, min(min(__double2hiint(pa), __double2hiint(da)), min(__double2hiint(pb), __double2hiint(db)))); rprod *= da * db;
        round2_nobook<2>(m, em, pa, da); round2_nobook<3>(m, em, pb, db);
        pdmin = min(pdmin, min(min(__double2hiint(pa), __double2hiint(da)), min(__double2hiint(pb), __double2hiint(db)))); rprod *= da * db;
        round2_nobook<4>(m, em, pa, da); round2_nobook<5>(m, em, pb, db);
        pdmin = min(pdmin, min(min(__double2hiint(pa), __double2hiint(da)), min(__double2hiint(pb), __double2hiint(db)))); rprod *= da * db;
    }
}

// m0: a 12 x 12 SPD matrix in the accumulator layout (lane (g, j), register r: entry (4 r + g, j)); every elimination starts from m0 plus a
// multiple of the previous result small enough not to change a bit of it (the dependency of step t on step t + 1)
template <int V>
__global__ __launch_bounds__(64) void bench(const double *m0g, double *out, unsigned long long *cyc, int reps) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    ElimMasks em; elim_masks(em, g, j);
    Masks1 mk; masks1(mk, g, j);
    d4 m0;
#pragma unroll
    for (int r = 0; r < 3; ++r) m0[r] = m0g[64 * r + l];
    m0[3] = 0.0;
    d4 m = m0;
    int pdmin = 1;
    double rprod = 1.0, carry = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int r = 0; r < 3; ++r) m[r] = fma(carry, 1e-300, m0[r]);
        m[3] = 0.0;
        rprod = 1.0;
        eliminate<V>(m, em, mk, pdmin, rprod);
        carry = m[0] + m[1] + m[2];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int r = 0; r < 3; ++r) out[(size_t)blockIdx.x * 256 + 64 * r + l] = m[r];
    if (l == 0) { out[(size_t)blockIdx.x * 256 + 192] = rprod; out[(size_t)blockIdx.x * 256 + 193] = (double)pdmin; cyc[blockIdx.x] = t1 - t0; }
}

int main() {
    const int n = 12;
    std::vector<double> M(n * n), img(192, 0.0);
    srand(5);
    std::vector<double> G(n * n);
    for (auto &x : G) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) for (int jj = 0; jj < n; ++jj) { double s = (i == jj) ? 2.0 : 0.0; for (int k = 0; k < n; ++k) s += 0.3 * G[i * n + k] * G[jj * n + k]; M[i * n + jj] = s; }
    for (int r = 0; r < 3; ++r) for (int l = 0; l < 64; ++l) { const int g = l >> 4, j = l & 15, i = 4 * r + g; if (j < 12) img[64 * r + l] = M[i * n + j]; }
    double *dm, *dout; unsigned long long *dc;
    const int NB = 2048;
    (void)hipMalloc(&dm, 192 * 8); (void)hipMalloc(&dout, (size_t)NB * 256 * 8); (void)hipMalloc(&dc, NB * 8);
    (void)hipMemcpy(dm, img.data(), 192 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    std::vector<double> ref(256);
    const char *names[4] = {"V0 2x2 block pivots (today)", "V1 1x1 pivots", "V2 1x1 pivots, next pivot ahead of the MFMA", "V3 2x2, bookkeeping every second round"};
    for (int v = 0; v < 4; ++v)
        for (int nb : {1024, 2048}) {                         // one / two wavefronts per SIMD (1,024 SIMDs)
            for (int rep = 0; rep < 2; ++rep) {
                switch (v) {
                    case 0: hipLaunchKernelGGL(bench<0>, dim3(nb), dim3(64), 0, 0, dm, dout, dc, reps); break;
                    case 1: hipLaunchKernelGGL(bench<1>, dim3(nb), dim3(64), 0, 0, dm, dout, dc, reps); break;
                    case 2: hipLaunchKernelGGL(bench<2>, dim3(nb), dim3(64), 0, 0, dm, dout, dc, reps); break;
                    default: hipLaunchKernelGGL(bench<3>, dim3(nb), dim3(64), 0, 0, dm, dout, dc, reps); break;
                }
                (void)hipDeviceSynchronize();
            }
            std::vector<unsigned long long> c(nb);
            std::vector<double> o(256);
            (void)hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0; unsigned long long mx = 0;
            for (auto x : c) { mean += (double)x; mx = x > mx ? x : mx; }
            mean /= nb;
            if (v == 0 && nb == 1024) ref = o;
            double err = 0, scale = 0;
            for (int e = 0; e < 192; ++e) { err = fmax(err, fabs(o[e] - ref[e])); scale = fmax(scale, fabs(ref[e])); }
            printf("%-48s %d wave(s) per SIMD: %7.1f cycles per elimination (slowest wave %7.1f) | max |diff| to V0's -inv(M) %.1e (scale %.1e) | det ratio %.3e pdmin>0 %d\n",
                   names[v], nb / 1024, mean / reps, (double)mx / reps, err, scale, o[192] / ref[192], o[193] > 0);
        }
    return 0;
}
