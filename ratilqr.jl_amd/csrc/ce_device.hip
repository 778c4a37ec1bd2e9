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

// get_positive_samples (:233-246) preceded by the head of step! (:259 iter_current += 1; :266-279 which (mu, sigma) to draw from).
// The sequential rule -- take standard normals in order, keep theta = mu + sigma z > 0 until num_samples are kept -- is replayed 1024
// stream elements at a time: an element's slot is the number of kept elements before it.
__global__ __launch_bounds__(CE_T) void ce_draw_kernel(CeDev *s, const double *__restrict__ z, long long z_avail, double *__restrict__ theta) {
#pragma clang fp contract(off)
    __shared__ int wsum[CE_T / 64];
    __shared__ double sh_mu, sh_sigma;
    __shared__ long long sh_pos, sh_newpos;
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
            sh_pos = s->zpos;
            sh_newpos = -1;
        }
    }
    __syncthreads();
    if (!sh_go) return;
    const double mu = sh_mu, sigma = sh_sigma;
    const int B = (int)s->num_samples;
    long long pos = sh_pos;
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
}

// The tail of step! (:291-334) on the costs of the batch: valid count, the redraw rules, theta_min / theta_max with the reference's
// if / elseif, elites under (isless(cost), index), mean and population standard deviation.
__global__ __launch_bounds__(CE_T) void ce_update_kernel(CeDev *s, const double *__restrict__ theta, const double *__restrict__ cost) {
#pragma clang fp contract(off)
    __shared__ double c_sh[CE_T], th_sh[CE_T];
    __shared__ int idx_sh[CE_T];
    __shared__ int wsum[CE_T / 64];
    __shared__ double wmin[CE_T / 64], wmax[CE_T / 64];
    __shared__ int sh_mode;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (s->error != 0 || s->draw_retry) return;                                            // (uniform: the draw of this slot did not happen)
    const int B = (int)s->num_samples, ne = (int)s->num_elite;
    const bool live = tid < B;
    const double c = live ? cost[tid] : INFINITY, th = live ? theta[tid] : 0.0;
    c_sh[tid] = c; th_sh[tid] = th;
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
    // inclusive prefix min inside the wave, then exclusive across waves
    double incl = tv;
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
    // elites: position of sample i in sort(by = cost) = number of samples strictly before it under (isless(cost), index): NaN last, ties
    // in input order (a stable sort)
    __syncthreads();
    int rank = 0;
    if (live) {
        const bool xn = c != c;
        for (int j = 0; j < B; ++j) {
            const double y = c_sh[j];
            const bool yn = y != y;
            bool before_me;                         // (y, j) < (c, tid) ?
            if (xn || yn) before_me = yn ? (xn && j < tid) : true;      // y not NaN, c NaN: y first; both NaN: by index; y NaN, c not: no
            else before_me = (y < c) || (!(c < y) && j < tid);
            rank += before_me ? 1 : 0;
        }
        if (rank < ne) idx_sh[rank] = tid;
    }
    __syncthreads();
    if (tid == 0) {
        double tmin = s->theta_min, tmax = s->theta_max;
#pragma unroll
        for (int w = 0; w < CE_T / 64; ++w) { tmin = fmin(tmin, wmin[w]); tmax = fmax(tmax, wmax[w]); }
        s->theta_min = tmin; s->theta_max = tmax;
        double sum = 0.0;
        for (int i = 0; i < ne; ++i) sum += th_sh[idx_sh[i]];
        const double mu_new = sum / (double)ne;                                            // :329
        double ss = 0.0;
        for (int i = 0; i < ne; ++i) { const double d = th_sh[idx_sh[i]] - mu_new; ss += d * d; }
        s->mu = mu_new;
        s->sigma = sqrt(ss / (double)ne);                                                  // :330-334 (population std)
        s->theta_opt = s->use_theta_max ? tmax : mu_new;                                   // solve! :375-382, after the last iteration
    }
}

void launch_ce_draw(CeDev *s, const double *z, long long z_avail, double *theta, hipStream_t st) {
    hipLaunchKernelGGL(ce_draw_kernel, dim3(1), dim3(CE_T), 0, st, s, z, z_avail, theta);
}
void launch_ce_update(CeDev *s, const double *theta, const double *cost, hipStream_t st) {
    hipLaunchKernelGGL(ce_update_kernel, dim3(1), dim3(CE_T), 0, st, s, theta, cost);
}
