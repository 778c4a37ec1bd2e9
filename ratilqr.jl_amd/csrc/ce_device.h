// ce_device.h -- device-resident state of one CrossEntropyBilevelOptimizationSolver solve! (ce_device.hip; host side: driver.cpp)
#pragma once
#include <hip/hip_runtime.h>

#define CE_ERR_DRY 1            /* the uploaded part of the standard-normal stream ran out inside a draw */
#define CE_DEV_MAX_B 1024       /* one workgroup draws / updates: batches up to 1024 samples */

struct CeDev {                  // mirrors rat_ce_solver (include/ratilqr.h) + the chain's own control words
    double mu_init, sigma_init, mu, sigma, theta_max, theta_min, lambda;
    long long iter_current, iter_max, num_samples, num_elite;
    long long n_solves, n_redraws;
    long long zpos;             // standard normals consumed so far (position in the device-resident stream)
    double theta_opt;           // use_theta_max ? theta_max : mu after the latest committed update (solve! :375-382)
    int use_theta_max;
    int redraw_pending;         // the latest update asked for a redraw: the next draw belongs to the same iteration (:293-298, :306)
    int this_is_redraw;         // the batch in flight is such a redraw (bookkeeping: n_redraws)
    int draw_retry;             // the latest draw ran out of normals before it had num_samples: nothing consumed, repeat it
    int error;                  // CE_ERR_*: every later kernel of the chain is a no-op
};

// one launch between two batches: update on the finished batch (do_update), draw of the next one (do_draw)
void launch_ce_step(CeDev *s, const double *z, long long z_avail, double *theta, const double *cost, int do_update, int do_draw, hipStream_t st);

// PETS step! bookkeeping on the device (ce_device.hip): control sequences from (mu_t, Sigma_t) and the elite / smoothed update
#define PETS_DEV_MAX_S 1024     /* one workgroup sorts the sample costs */
// one launch between two rollout launches: elites + smoothed update on the finished rollouts (do_update), the next control sequences (do_sample)
struct PetsStage3 { const double *src[3]; double *dst[3]; long n[3]; int *zero_word; };      // three copies (and one word zeroed) in one launch
void launch_pets_stage3(const PetsStage3 &a, hipStream_t st);
void launch_pets_step(double *mu, double *Sigma, double *controls, const double *cost, long S, int ne, int N, int m, double sf, const double *zc,
                      unsigned long long seed, int it, int do_update, int do_sample, int *err, hipStream_t st);
