// ce_device.hip -- the bookkeeping half of the Cross-Entropy step! on the device (cross_entropy_bilevel_optimization.jl:233-246, 252-335).
//
// rat_ce_solve used to make one host round trip per CE iteration: draw theta on the host, upload, batch of solves, download the costs,
// count / sort / update on the host.  Here the draw (get_positive_samples :233-246) and the update (:291-334) are two one-workgroup
// kernels working on a CeDev record in HBM, so that a whole solve! is ONE enqueue chain
//     [draw -> batch of iLEQG solves -> update] x iter_max -> final solve at theta_opt -> one copy back, one host wait
// with the host off the critical path (it only generates / uploads standard normals ahead of their use).  The arithmetic is the host
// code's (driver.cpp rat_ce_get_positive_samples / rat_ce_update, which stay for rat_ce_step and the multi-device solver) operation for
// operation -- same operand order, no contraction -- so mu, sigma, theta_min / theta_max and theta_opt are bit-identical to the host path
// and to the oracle's (tests/test_gpu_ce.py).  One workgroup of 1024 threads: batches up to 1024 samples (the CE default is 10; BASELINE: 1024).
#include <hip/hip_runtime.h>
#include <math.h>

#include "ce_device.h"
#include "rat_normal.h"

#define CE_T 1024

// block-wide exclusive prefix sum of a 0/1 flag and its total (1024 threads = 16 wavefronts)
__device__ __forceinline__ int block_excl_scan(const bool flag, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                                   // (wsum is reused across calls)
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CE_T / 64; ++w) { const int c = wsum[w]; if (w < wave) base += c; tot += c; }
    total = tot;
    return base + in_wave;
}

// The record is 160 bytes: every kernel works on a copy in LDS (one coalesced load / store instead of a chain of dependent global
// round trips by one lane) and writes it back at its end.
#define CE_WORDS ((int)(sizeof(CeDev) / 8))
static_assert(sizeof(CeDev) % 8 == 0, "CeDev is copied in 8-byte words");
__device__ __forceinline__ void rec_load(CeDev *sc, const CeDev *g) {
    if (threadIdx.x < CE_WORDS) reinterpret_cast<unsigned long long *>(sc)[threadIdx.x] = reinterpret_cast<const unsigned long long *>(g)[threadIdx.x];
    __syncthreads();
}
__device__ __forceinline__ void rec_store(CeDev *g, const CeDev *sc) {
    __syncthreads();
    if (threadIdx.x < CE_WORDS) reinterpret_cast<unsigned long long *>(g)[threadIdx.x] = reinterpret_cast<const unsigned long long *>(sc)[threadIdx.x];
}

// get_positive_samples (:233-246) preceded by the head of step! (:259 iter_current += 1; :266-279 which (mu, sigma) to draw from).
// The sequential rule -- take standard normals in order, keep theta = mu + sigma z > 0 until num_samples are kept -- is replayed 1024
// stream elements at a time: an element's slot is the number of kept elements before it.
__device__ __forceinline__ void ce_draw(CeDev *const s, const double *__restrict__ z, const long long z_avail, double *__restrict__ theta, int *wsum) {
#pragma clang fp contract(off)
    __shared__ double sh_mu, sh_sigma;
    __shared__ long long sh_newpos;
    __shared__ int sh_go;
    const int tid = threadIdx.x;
    if (tid == 0) {
        sh_go = (s->error == 0 && (s->iter_current < s->iter_max || s->redraw_pending || s->draw_retry)) ? 1 : 0;
        if (sh_go) {
            if (!s->redraw_pending && !s->draw_retry) s->iter_current += 1;                // step! :259
            s->this_is_redraw = s->redraw_pending;
            const bool first = s->iter_current == 1;                                       // :266-279
            sh_mu = first ? s->mu_init : s->mu;
            sh_sigma = first ? s->sigma_init : s->sigma;
            sh_newpos = -1;
        }
    }
    __syncthreads();
    if (!sh_go) return;
    const double mu = sh_mu, sigma = sh_sigma;
    const int B = (int)s->num_samples;
    long long pos = s->zpos;
    int count = 0;
    bool dry = false;
    while (count < B) {
        const long long idx = pos + tid;
        const bool valid = idx < z_avail;
        const double th = valid ? mu + sigma * z[idx] : 0.0;                               // rand(rng, Normal(mu, sigma))
        const bool keep = valid && th > 0.0;
        int total;
        const int rank = count + block_excl_scan(keep, wsum, total);
        if (keep && rank < B) theta[rank] = th;
        if (keep && rank == B - 1) sh_newpos = idx + 1;                                    // the stream position after the last kept draw
        count += total;
        pos += CE_T;
        if (count < B && pos >= z_avail) { dry = true; break; }                            // (uniform: every thread sees the same count / pos)
    }
    __syncthreads();
    if (tid == 0) {
        if (dry) { s->error = CE_ERR_DRY; s->draw_retry = 1; }                             // nothing consumed: the host tops the stream up and re-enqueues
        else { s->zpos = sh_newpos; s->draw_retry = 0; }
    }
    __syncthreads();
}

// sort(by = cost) orders by isless on the costs -- Julia's total order on floats: NaN after everything, -0.0 BEFORE +0.0 (isless(-0.0, 0.0)
// is true) -- with ties in input order (stable).  The cost is mapped once to an unsigned key with the same order (sign-magnitude -> biased:
// the key of -0.0 is one below that of +0.0; every NaN -> the largest key), so that a
// compare-exchange of the network is three integer comparisons and no branch (the comparator written on doubles, with its NaN cases,
// compiled to nested divergent branches: 20 us per update instead of ~6).
__device__ __forceinline__ unsigned long long elite_key(double c) {
    if (c != c) return ~0ull;
    const unsigned long long b = (unsigned long long)__double_as_longlong(c);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// The tail of step! (:291-334) on the costs of the batch: valid count, the redraw rules, theta_min / theta_max with the reference's
// if / elseif, elites under (isless(cost), index), mean and population standard deviation.
__device__ __forceinline__ void ce_update(CeDev *const s, const double *__restrict__ theta, const double *__restrict__ cost, int *wsum) {
#pragma clang fp contract(off)
    __shared__ double c_sh[CE_T], th_sh[CE_T], el_sh[CE_T];
    __shared__ int idx_sh[CE_T];
    __shared__ double wmin[CE_T / 64], wmax[CE_T / 64];
    __shared__ int sh_mode;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (s->error != 0 || s->draw_retry) return;                                            // (uniform: the draw of this slot did not happen)
    const int B = (int)s->num_samples, ne = (int)s->num_elite;
    const bool live = tid < B;
    const double c = live ? cost[tid] : NAN;                                                     // (pad slots: NaN cost, index >= B: after every sample)
    const double th = live ? theta[tid] : 0.0;
    th_sh[tid] = th;
    const bool isinf_c = live && (c == INFINITY || c == -INFINITY);                        // isinf(cost) :291 (a NaN cost is not infinite)
    int num_inf;
    (void)block_excl_scan(isinf_c, wsum, num_inf);
    if (tid == 0) {
        const long long num_valid = (long long)B - num_inf;
        const double thresh = fmax((double)ne, (double)B * s->lambda);
        int mode;                                                                          // 0 commit, 1 redraw
        s->n_solves += B;
        if (s->this_is_redraw) s->n_redraws += 1;
        if (s->iter_current == 1 && (double)num_valid < thresh) {                          // :293-298
            s->mu_init *= s->lambda; s->sigma_init *= s->lambda;
            mode = 1;
        } else if (s->iter_current == 1 && num_valid == B) {                               // :299-305
            s->mu_init /= s->lambda; s->sigma_init /= s->lambda;
            mode = 0;
        } else if ((double)num_valid >= thresh) {                                          // :306
            mode = 0;
        } else mode = 1;                                                                   // redraw with unchanged parameters
        s->redraw_pending = mode;
        sh_mode = mode;
    }
    __syncthreads();
    if (sh_mode) return;
    // theta_min / theta_max (:314-324): `if theta < theta_min ... elseif theta > theta_max` -- a sample that lowers the running minimum
    // is not looked at for the maximum.  theta_min_out = min(theta_min_in, valid thetas); a valid sample i counts towards theta_max iff
    // NOT theta_i < (running minimum before i) = min(theta_min_in, valid thetas before i): an exclusive prefix minimum.
    const bool valid = live && !isinf_c;
    const double tv = valid ? th : INFINITY;
    double incl = tv;                                   // inclusive prefix minimum inside the wave, then exclusive across waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(incl, off, 64); if (lane >= off) incl = fmin(incl, o); }
    if (lane == 63) wmin[wave] = incl;
    __syncthreads();
    double before = s->theta_min;                                                          // running minimum before this thread's sample
#pragma unroll
    for (int w = 0; w < CE_T / 64; ++w) if (w < wave) before = fmin(before, wmin[w]);
    { const double up = __shfl_up(incl, 1, 64); if (lane > 0) before = fmin(before, up); }
    const bool for_max = valid && !(th < before);
    double mx = for_max ? th : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) wmax[wave] = mx;
    // elites: sort(by = cost), stable = the strict total order (isless(cost), index): a bitonic network over the 1024 slots of the
    // workgroup, one element per thread in registers -- partners inside a wavefront (distance < 64) exchange by lane shuffles, the ten
    // stages at distance >= 64 through LDS.  (An O(B^2) rank count on ONE compute unit took 85 us, all 55 stages through LDS 7 us.)
    int id = tid;
    unsigned long long key = elite_key(c);
    unsigned long long *const k_sh = reinterpret_cast<unsigned long long *>(c_sh);
#pragma unroll
    for (int k = 2; k <= CE_T; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long ok; int oi;
            if (j >= 64) {
                __syncthreads();
                k_sh[tid] = key; idx_sh[tid] = id;
                __syncthreads();
                ok = k_sh[tid ^ j]; oi = idx_sh[tid ^ j];
            } else {
                ok = (unsigned long long)__shfl_xor((long long)key, j, 64); oi = __shfl_xor(id, j, 64);
            }
            // the lower position of a pair keeps the smaller element in an ascending block, the larger one in a descending block
            const bool want_min = ((tid & j) == 0) == ((tid & k) == 0);
            const bool mine_first = (key < ok) | ((key == ok) & (id < oi));
            const bool take = want_min != mine_first;
            key = take ? ok : key;
            id = take ? oi : id;
        }
    }
    __syncthreads();
    if (tid < ne) el_sh[tid] = th_sh[id];                      // the elites' thetas in sorted order (position tid of the sorted sequence)
    __syncthreads();
    if (tid == 0) {
        double tmin = s->theta_min, tmax = s->theta_max;
#pragma unroll
        for (int w = 0; w < CE_T / 64; ++w) { tmin = fmin(tmin, wmin[w]); tmax = fmax(tmax, wmax[w]); }
        s->theta_min = tmin; s->theta_max = tmax;
        double sum = 0.0;
#pragma unroll 8
        for (int i = 0; i < ne; ++i) sum += el_sh[i];
        const double mu_new = sum / (double)ne;                                            // :329
        double ss = 0.0;
#pragma unroll 8
        for (int i = 0; i < ne; ++i) { const double d = el_sh[i] - mu_new; ss += d * d; }
        s->mu = mu_new;
        s->sigma = sqrt(ss / (double)ne);                                                  // :330-334 (population std)
        s->theta_opt = s->use_theta_max ? tmax : mu_new;                                   // solve! :375-382, after the last iteration
    }
    __syncthreads();
}

// One bookkeeping launch between two batches: the update on the batch that has just finished (do_update), then the draw of the next one
// (do_draw) -- the draw finds the iteration finished, a redraw pending or an error in the record and acts accordingly.
__global__ __launch_bounds__(CE_T) void ce_step_kernel(CeDev *sg, const double *__restrict__ z, long long z_avail, double *__restrict__ theta,
                                                       const double *__restrict__ cost, int do_update, int do_draw) {
    __shared__ int wsum[CE_T / 64];
    __shared__ CeDev sc;
    rec_load(&sc, sg);
    if (do_update) ce_update(&sc, theta, cost, wsum);
    __syncthreads();
    if (do_draw) ce_draw(&sc, z, z_avail, theta, wsum);
    rec_store(sg, &sc);
}

void launch_ce_step(CeDev *s, const double *z, long long z_avail, double *theta, const double *cost, int do_update, int do_draw, hipStream_t st) {
    hipLaunchKernelGGL(ce_step_kernel, dim3(1), dim3(CE_T), 0, st, s, z, z_avail, theta, cost, do_update, do_draw);
}


// =====================================================================================================================================
// PETS (CrossEntropyDirectOptimizationSolver, pets.jl:159-245): the bookkeeping of step! on the device, so that solve! (:270-281) is ONE
// enqueue chain  [sample control sequences -> rollouts -> elites + smoothed update] x iter_max -> one copy back, one host wait.
// The arithmetic is the host code's (driver.cpp rat_pets_sample_controls / rat_pets_update) operation for operation, no contraction:
// mu and Sigma come out bit-identical to the host loop on an injected stream of control normals (tests/test_gpu_pets.py).
// =====================================================================================================================================
// Philox4x32-10 (as in kernels.hip) for control normals drawn on the device (zc == nullptr)
__device__ __forceinline__ void pd_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double pd_u01(unsigned hi, unsigned lo) { return (double)((((unsigned long long)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

// ONE bookkeeping launch between two rollout launches.  Round 6: a GRID of workgroups, each owning a slice of the horizon (one time step per
// workgroup up to 64 of them) -- everything below is independent per time step except the ranking of the sample costs, which every
// workgroup repeats for itself (1024 threads; the sort runs over CT slots, CT = the power of two >= S, at least 64 -- a batch of 100
// control samples needs 28 of the 55 stages of the 1024-slot network, one of them through LDS).  A workgroup reads and writes the controls
// of ITS time steps only (reads before writes, behind its own barrier), so the update and the next draw stay in place:
//   do_update: get_elite_samples + compute_new_distribution (pets.jl:159-191) on the sample costs of the finished rollouts -- elites by the
//              bitonic network of ce_update under (isless(cost), index), over CT slots only; then per time step of the slice: the elites'
//              controls pulled into LDS by all threads at once, a thread per control a: mean and unbiased variance over the elites in
//              sorted order (sequential sums: the reference's order), smoothed update of mu_t and of the diagonal covariance;
//   do_sample: the next iteration's control sequences, controls[ii][t][0..3] = mu_t + chol(Sigma_t) z  (rand(rng, MvNormal(mu_t, Sigma_t)),
//              :206-216), padded to four controls: the slice's covariances are factorised by a thread per time step (m <= 4), then its
//              (ii, t) pairs in strides of 1024.  zc: injected standard normals [S][N][m] of that iteration, or nullptr: Philox keyed by
//              (seed, iteration).
template <int CT>
__global__ __launch_bounds__(CE_T) void pets_step_kernel(double *__restrict__ mu, double *__restrict__ Sigma, double *__restrict__ controls,
                                                       const double *__restrict__ cost, long S, int ne, int N, int m, double sf,
                                                       const double *__restrict__ zc, unsigned long long seed, int it, int do_update, int do_sample, int *err) {
#pragma clang fp contract(off)
    __shared__ unsigned long long k_sh[CT];
    __shared__ int idx_sh[CT];
    extern __shared__ double Lsh[];                          // [N][16] Cholesky factors | [ne][4] the elites' controls of one time step
    double *const v_sh = Lsh + (size_t)N * 16;
    const int tid = threadIdx.x;
    const int t_lo = (int)((long)blockIdx.x * N / gridDim.x), t_hi = (int)((long)(blockIdx.x + 1) * N / gridDim.x), nt = t_hi - t_lo;
    if (do_update) {
        const bool sorter = tid < CT;                        // (whole wavefronts: CT is a multiple of 64)
        const double c = (tid < S) ? cost[tid] : NAN;        // (pad slots: NaN cost, index >= S: after every sample)
        int id = tid;
        unsigned long long key = elite_key(c);
#pragma unroll
        for (int k = 2; k <= CT; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long ok = key; int oi = id;
                if (j >= 64) {
                    __syncthreads();
                    if (sorter) { k_sh[tid] = key; idx_sh[tid] = id; }
                    __syncthreads();
                    if (sorter) { ok = k_sh[tid ^ j]; oi = idx_sh[tid ^ j]; }
                } else if (sorter) {
                    ok = (unsigned long long)__shfl_xor((long long)key, j, 64); oi = __shfl_xor(id, j, 64);
                }
                const bool want_min = ((tid & j) == 0) == ((tid & k) == 0);
                const bool mine_first = (key < ok) | ((key == ok) & (id < oi));
                const bool take = sorter && (want_min != mine_first);
                key = take ? ok : key;
                id = take ? oi : id;
            }
        }
        __syncthreads();
        if (sorter) idx_sh[tid] = id;                        // position tid of the sorted sequence
        __syncthreads();
        for (int t = t_lo; t < t_hi; ++t) {
            for (int q = tid; q < ne * 4; q += CE_T) v_sh[q] = controls[((long)idx_sh[q >> 2] * N + t) * 4 + (q & 3)];
            __syncthreads();
            if (tid < m) {
                const int a = tid;
                double mean = 0.0;
                for (int e0 = 0; e0 < ne; ++e0) mean += v_sh[e0 * 4 + a];
                mean /= (double)ne;                                                    // :183
                double var = 0.0;
                for (int e0 = 0; e0 < ne; ++e0) { const double d = v_sh[e0 * 4 + a] - mean; var += d * d; }
                var /= (double)(ne - 1);                                               // var = unbiased :184
                mu[t * m + a] = (1.0 - sf) * mean + sf * mu[t * m + a];                 // :186
                for (int b = 0; b < m; ++b) {                                          // Diagonal(var) :184, smoothing :187
                    double *Sg = &Sigma[(long)t * m * m + a + m * b];
                    *Sg = (1.0 - sf) * (a == b ? var : 0.0) + sf * *Sg;
                }
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();                                     // (the sampling below reads the mu / Sigma just written, and overwrites the controls)
    }
    if (!do_sample) return;
    for (int t = t_lo + tid; t < t_hi; t += CE_T) {          // host_chol_lower, column-major m x m
        const double *A = Sigma + (long)t * m * m;
        double *Lo = Lsh + t * 16;
        for (int q = 0; q < 16; ++q) Lo[q] = 0.0;
        bool ok = true;
        for (int j = 0; j < m && ok; ++j) {
            double d = A[j + m * j];
            for (int k = 0; k < j; ++k) d -= Lo[j + m * k] * Lo[j + m * k];
            if (!(d > 0.0)) { ok = false; break; }
            Lo[j + m * j] = sqrt(d);
            for (int i = j + 1; i < m; ++i) {
                double v = A[i + m * j];
                for (int k = 0; k < j; ++k) v -= Lo[i + m * k] * Lo[j + m * k];
                Lo[i + m * j] = v / Lo[j + m * j];
            }
        }
        if (!ok) atomicExch(err, 1);                         // Sigma_t is not positive definite (MvNormal would throw)
    }
    __syncthreads();
    for (long e = tid; e < S * nt; e += CE_T) {
        const long ii = e / nt;
        const int t = t_lo + (int)(e - ii * nt);
        double z[4] = {0.0, 0.0, 0.0, 0.0};
        if (zc) {
            for (int b = 0; b < m; ++b) z[b] = zc[(ii * N + t) * m + b];
        } else {
            unsigned r[4];
            pd_philox((unsigned)ii, (unsigned)(ii >> 32), (unsigned)t, 0x50455453u, (unsigned)seed ^ 0xC0117201u, (unsigned)(seed >> 32) + (unsigned)it, r);
            ratn_box_muller(pd_u01(r[0], r[1]), pd_u01(r[2], r[3]), &z[0], &z[1]);
            if (m > 2) {
                pd_philox((unsigned)ii, (unsigned)(ii >> 32), (unsigned)t, 0x50455454u, (unsigned)seed ^ 0xC0117201u, (unsigned)(seed >> 32) + (unsigned)it, r);
                ratn_box_muller(pd_u01(r[0], r[1]), pd_u01(r[2], r[3]), &z[2], &z[3]);
            }
        }
        const double *Lt = Lsh + t * 16;
        double out[4] = {0.0, 0.0, 0.0, 0.0};
        for (int a = 0; a < m; ++a) {
            double v = mu[t * m + a];
            for (int b = 0; b <= a; ++b) v += Lt[a + m * b] * z[b];
            out[a] = v;
        }
        double *c = controls + (ii * N + t) * 4;
        c[0] = out[0]; c[1] = out[1]; c[2] = out[2]; c[3] = out[3];
    }
}

// Up to three copies in ONE launch (x0 | mu | Sigma in from the pinned area; mu | Sigma | error word back out): a launch behind a launch
// costs ~2-4 us on the stream, and the device-resident loop is ~20 of them around 0.2 ms of rollouts.
__global__ __launch_bounds__(256) void pets_stage3_kernel(PetsStage3 a) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && a.zero_word) *a.zero_word = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (i >= 0 && i < a.n[k]) a.dst[k][i] = a.src[k][i];
        i -= a.n[k];
    }
}
void launch_pets_stage3(const PetsStage3 &a, hipStream_t st) {
    const long total = a.n[0] + a.n[1] + a.n[2];
    if (total <= 0) return;
    hipLaunchKernelGGL(pets_stage3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

void launch_pets_step(double *mu, double *Sigma, double *controls, const double *cost, long S, int ne, int N, int m, double sf, const double *zc,
                      unsigned long long seed, int it, int do_update, int do_sample, int *err, hipStream_t st) {
    const size_t lds = ((size_t)N * 16 + (size_t)ne * 4) * sizeof(double);
    const int nblk = N < 64 ? N : 64;                        // one time step per workgroup up to 64 workgroups, slices beyond
    // beyond 64 KB of LDS (static ~12 KB + 128 B per time step: horizons above ~400) a launch needs the attribute raised first, as
    // launch_wide_solve does; a failure here surfaces through hipGetLastError at the caller's check like a failed launch
#define PSTEP(CT) do { if (lds + 16 * 1024 > 64 * 1024) \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pets_step_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pets_step_kernel<CT>), dim3(nblk), dim3(CE_T), lds, st, mu, Sigma, controls, cost, S, ne, N, m, sf, zc, seed, it, do_update, do_sample, err); } while (0)
    if (S <= 64) PSTEP(64); else if (S <= 128) PSTEP(128); else if (S <= 256) PSTEP(256); else if (S <= 512) PSTEP(512); else PSTEP(1024);
#undef PSTEP
}
