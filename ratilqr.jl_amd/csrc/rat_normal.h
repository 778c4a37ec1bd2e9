// rat_normal.h -- the Box-Muller transform of the device noise generators (PETS rollouts, rat_rollout_noisy), written out.
//
//     (z0, z1) = sqrt(-2 ln(1 - u1)) (cos 2 pi u2, sin 2 pi u2),   u1, u2 uniform in [0, 1) on a 2^-53 grid.
//
// The vendor's log / sqrt / sincospi are general (denormals, infinities, every argument range) and cost ~125 fp64 operations per transform;
// the three transforms of a step pair were a third of a PETS step.  Here the arguments are what they are -- 1 - u1 in [2^-53, 1], the
// radicand in [0, 74], the angle 2 u2 in [0, 2) -- so: ln by fdlibm's e_log.c reduction and degree-7 polynomial (m in [sqrt(1/2), sqrt(2)),
// s = f / (2 + f); < 1 ulp); the root by a reciprocal-root seed and two Newton steps; sin / cos of pi t by exact reduction to
// r = t - q / 2 in [-1/4, 1/4] and the Taylor polynomials of sin(pi r), cos(pi r) to r^17 / r^16 (truncation < 1e-17).  ~75 operations,
// results within 2 ulp of the library's (tests/test_cpu_normal.py holds the header against libm on the CPU through oracle/normal_check.c;
// tests/test_gpu_pets.py holds the device's costs against a host restatement of Philox + this transform).
// Plain C: the CPU check includes this very file.
//
// fdlibm's notice, preserved as its licence asks (ratn_log: e_log.c's reduction, polynomial and constants):
// ====================================================
// Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
//
// Developed at SunSoft, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice
// is preserved.
// ====================================================
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define RATN_FN __device__ __forceinline__
#define RATN_RSQ(x) __builtin_amdgcn_rsq(x)
#define RATN_RCP(x) __builtin_amdgcn_rcp(x)
#elif defined(__HIPCC__)
#define RATN_FN __host__ __device__ inline
#define RATN_RSQ(x) (1.0 / sqrt(x))
#define RATN_RCP(x) (1.0 / (x))
#else
#define RATN_FN static inline
#define RATN_RSQ(x) (1.0 / sqrt(x))
#define RATN_RCP(x) (1.0 / (x))
#endif

// ln(v), v in [2^-53, 1] (any positive normal number works)
RATN_FN double ratn_log(const double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1022;                    // v = m 2^e, m in [1/2, 1)
    b = (b & 0x800fffffffffffffull) | 0x3fe0000000000000ull;
    double m;
    memcpy(&m, &b, 8);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }      // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0, d = 2.0 + f;
    double inv = RATN_RCP(d);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
    const double s = f * inv, z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// sqrt(a), a in [0, 1e3]
RATN_FN double ratn_sqrt(const double a) {
    double y = RATN_RSQ(a);
    const double h = 0.5 * a;
    y = y * fma(-h, y * y, 1.5);
    y = y * fma(-h, y * y, 1.5);
    double s = a * y;
    s = fma(fma(-s, s, a), 0.5 * y, s);
    return (a > 0.0) ? s : 0.0;
}

// sin(pi t), cos(pi t), t in [0, 2):  t = q / 2 + r, q = 0 .. 4, r in [-1/4, 1/4];  coefficients (-1)^k pi^(2k+1) / (2k+1)!, (-1)^k pi^(2k) / (2k)!
RATN_FN void ratn_sincospi(const double t, double *sn, double *cs) {
    const double qf = rint(t * 2.0);
    const double r = fma(-0.5, qf, t);                          // exact
    const double r2 = r * r;
    double sp = 7.95205400147551261e-07;
    sp = fma(sp, r2, -2.19153534478302173e-05);
    sp = fma(sp, r2, 4.66302805767612554e-04);
    sp = fma(sp, r2, -7.37043094571435044e-03);
    sp = fma(sp, r2, 8.21458866111282326e-02);
    sp = fma(sp, r2, -5.99264529320792105e-01);
    sp = fma(sp, r2, 2.55016403987734552e+00);
    sp = fma(sp, r2, -5.16771278004997026e+00);
    sp = fma(sp, r2, 3.14159265358979312e+00);
    sp *= r;
    double cp = 4.30306958703294729e-06;
    cp = fma(cp, r2, -1.04638104924845705e-04);
    cp = fma(cp, r2, 1.92957430940392314e-03);
    cp = fma(cp, r2, -2.58068913900140612e-02);
    cp = fma(cp, r2, 2.35330630358893206e-01);
    cp = fma(cp, r2, -1.33526276885458950e+00);
    cp = fma(cp, r2, 4.05871212641676848e+00);
    cp = fma(cp, r2, -4.93480220054467900e+00);
    cp = fma(cp, r2, 1.0);
    const int q = (int)qf & 3;                                  // sin(q pi / 2 + x), cos(q pi / 2 + x)
    const double s1 = (q & 1) ? cp : sp, c1 = (q & 1) ? sp : cp;
    *sn = (q & 2) ? -s1 : s1;
    *cs = (q == 1 || q == 2) ? -c1 : c1;
}

// both normals of one transform
RATN_FN void ratn_box_muller(const double u1, const double u2, double *z0, double *z1) {
    const double rad = ratn_sqrt(-2.0 * ratn_log(1.0 - u1));
    double sn, cs;
    ratn_sincospi(2.0 * u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}
