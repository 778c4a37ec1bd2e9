// rat_pow.h -- x^y for doubles as the reference computes it, bit for bit, on the device.
//
// Reference arithmetic.  The power-law family (test/ileqg_test.jl:151-155: f = x.^1.3 + u.^1.5, c = sum(x.^2.5 + u.^2.5)) evaluates
// Float64 ^ Float64, which Julia 1.5 (docs/Manifest: the reference's Julia) lowers to the `pow` of its bundled openlibm -- the FreeBSD
// msun e_pow.c (Sun Microsystems' fdlibm __ieee754_pow, error < 0.70 ulp), NOT a correctly rounded power and NOT the device math
// library's pow (ocml: different polynomial, ~1-2 ulp away on ~1 % of arguments, which is what made long power-law iterations drift
// from the oracle).  This header restates that published algorithm in plain operations -- every product, sum and
// quotient is a single IEEE operation in the order of the algorithm's description, no fused multiply-add (the splitting into high and
// low words relies on exact products of truncated operands) -- so that the device and the CPU oracle's own restatement
// (oracle/fdlibm_pow.h) produce identical bits; tests/test_cpu_pow.py compiles both with gcc and compares them with each other on 10^6
// arguments and with the host libm (<= 1 ulp: two different < 1 ulp algorithms), tests/test_gpu_pow.py compares device bits with oracle bits.
//
// The algorithm (fdlibm, "e_pow.c 1.5 04/04/22"):
//   1. log2(x) = n + log2(ax / bp) + log2(bp), bp in {1, 1.5} chosen from the mantissa; s = (ax - bp) / (ax + bp) split into s_h + s_l;
//      log(ax/bp) = 2 s + 2/3 s^3 + s^5 R(s^2) with the degree-6 polynomial L1..L6; result as t1 + t2 with t1 truncated to 32 bits.
//   2. y * log2(x) = (y1 + y2)(t1 + t2) with y1 = y truncated: p_h + p_l, overflow / underflow decided here.
//   3. 2^(p_h + p_l): n = round(p), 2^r by exp(r ln 2) = 1 - ((r t1') / (t1' - 2) - (w + r w)) - r with the degree-5 Remez P1..P5.
//   Special cases (y = 0, +-1, 2, 0.5, +-inf; x = 0, +-1, +-inf, x < 0 with integer / non-integer y, NaNs) exactly as fdlibm.
//
// fdlibm's notice, preserved as its licence asks (the algorithm, its step order and its published constants are fdlibm's):
// ====================================================
// Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
//
// Developed at SunSoft, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice
// is preserved.
// ====================================================
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RAT_POW_FN __host__ __device__ inline
#else
#define RAT_POW_FN static inline
#endif

RAT_POW_FN uint64_t ratpow_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
RAT_POW_FN double ratpow_from(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
RAT_POW_FN int32_t ratpow_hi(double x) { return (int32_t)(ratpow_bits(x) >> 32); }
RAT_POW_FN uint32_t ratpow_lo(double x) { return (uint32_t)ratpow_bits(x); }
RAT_POW_FN double ratpow_words(int32_t hi, uint32_t lo) { return ratpow_from(((uint64_t)(uint32_t)hi << 32) | lo); }
RAT_POW_FN double ratpow_trunc(double x) { return ratpow_from(ratpow_bits(x) & 0xFFFFFFFF00000000ull); }      /* low word <- 0 */
RAT_POW_FN double ratpow_scalbn(double z, int n) {          /* z 2^n for a result in the subnormal range (z in [1/2, 2), -1100 < n < 0): exact */
    const double down = ratpow_words((int32_t)((uint32_t)(0x3ff - 1000) << 20), 0);      /* scaling by 2^-1000, then ONE rounding multiply */
    if (n < -1000) { z *= down; n += 1000; }
    return z * ratpow_words((int32_t)((uint32_t)(0x3ff + n) << 20), 0);
}

RAT_POW_FN double rat_pow(double x, double y) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double bp1 = 1.5;
    const double dp_h1 = 5.84962487220764160156e-01, dp_l1 = 1.35003920212974897128e-08;
    const double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300;
    const double L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                 L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09;
    const double ovt = 8.0085662595372944372e-17;
    const double cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01, cp_l = -7.02846165095275826516e-09;
    const double ivln2 = 1.44269504088896338700e+00, ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;

    int32_t hx = ratpow_hi(x), hy = ratpow_hi(y);
    const uint32_t lx = ratpow_lo(x), ly = ratpow_lo(y);
    int32_t ix = hx & 0x7fffffff;
    const int32_t iy = hy & 0x7fffffff;

    if ((iy | (int32_t)ly) == 0) return 1.0;                                   /* x^0 = 1, even for NaN */
    if (hx == 0x3ff00000 && lx == 0) return 1.0;                               /* 1^y = 1, even for NaN */
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lx != 0) || iy > 0x7ff00000 || (iy == 0x7ff00000 && ly != 0)) return (x + 0.0) + (y + 0.0);

    /* x < 0: is y an odd integer (1), an even integer (2), or not an integer (0)? */
    int yisint = 0;
    if (hx < 0) {
        if (iy >= 0x43400000) yisint = 2;
        else if (iy >= 0x3ff00000) {
            const int k = (iy >> 20) - 0x3ff;
            if (k > 20) {
                const uint32_t j = ly >> (52 - k);
                if ((j << (52 - k)) == ly) yisint = 2 - (int)(j & 1);
            } else if (ly == 0) {
                const int32_t j = iy >> (20 - k);
                if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
            }
        }
    }
    if (ly == 0) {                                                             /* special values of y */
        if (iy == 0x7ff00000) {
            if (((ix - 0x3ff00000) | (int32_t)lx) == 0) return 1.0;            /* (-1)^+-inf = 1 */
            else if (ix >= 0x3ff00000) return (hy >= 0) ? y : 0.0;
            else return (hy < 0) ? -y : 0.0;
        }
        if (iy == 0x3ff00000) return (hy < 0) ? 1.0 / x : x;
        if (hy == 0x40000000) return x * x;
        if (hy == 0x3fe00000 && hx >= 0) return __builtin_sqrt(x);
    }
    double ax = __builtin_fabs(x);
    if (lx == 0 && (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000)) {        /* x is +-0, +-inf, +-1 */
        double z = ax;
        if (hy < 0) z = 1.0 / z;
        if (hx < 0) {
            if (((ix - 0x3ff00000) | yisint) == 0) z = (z - z) / (z - z);      /* (-1)^non-integer: NaN */
            else if (yisint == 1) z = -z;
        }
        return z;
    }
    int32_t n = (int32_t)(((uint32_t)hx >> 31)) - 1;                           /* 0 for x < 0, -1 otherwise */
    if ((n | yisint) == 0) return (x - x) / (x - x);                           /* (x < 0)^non-integer: NaN */
    double sgn = 1.0;
    if ((n | (yisint - 1)) == 0) sgn = -1.0;                                   /* (x < 0)^odd */

    double t1, t2;
    if (iy > 0x41e00000) {                                                     /* |y| > 2^31 */
        if (iy > 0x43f00000) {                                                 /* |y| > 2^64: over / underflow */
            if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
            if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
        }
        if (ix < 0x3fefffff) return (hy < 0) ? sgn * huge * huge : sgn * tiny * tiny;
        if (ix > 0x3ff00000) return (hy > 0) ? sgn * huge * huge : sgn * tiny * tiny;
        /* |1 - x| <= 2^-20: log(x) by x - x^2/2 + x^3/3 - x^4/4 */
        const double t = ax - 1.0;
        const double w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
        const double u = ivln2_h * t;
        const double v = t * ivln2_l - w * ivln2;
        t1 = ratpow_trunc(u + v);
        t2 = v - (t1 - u);
    } else {
        n = 0;
        if (ix < 0x00100000) { ax *= two53; n -= 53; ix = ratpow_hi(ax); }     /* subnormal x */
        n += (ix >> 20) - 0x3ff;
        const int32_t j = ix & 0x000fffff;
        int k;
        ix = j | 0x3ff00000;
        if (j <= 0x3988E) k = 0;                                               /* |x| < sqrt(3/2) */
        else if (j < 0xBB67A) k = 1;                                           /* |x| < sqrt(3)   */
        else { k = 0; n += 1; ix -= 0x00100000; }
        ax = ratpow_words(ix, ratpow_lo(ax));
        const double bpk = k ? bp1 : 1.0, dphk = k ? dp_h1 : 0.0, dplk = k ? dp_l1 : 0.0;
        /* ss = s_h + s_l = (x - bp) / (x + bp) */
        double u = ax - bpk;
        double v = 1.0 / (ax + bpk);
        const double ss = u * v;
        const double s_h = ratpow_trunc(ss);
        double t_h = ratpow_words(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18), 0);      /* ax + bp, high part */
        double t_l = ax - (t_h - bpk);
        const double s_l = v * ((u - s_h * t_h) - s_h * t_l);
        /* log(ax) */
        double s2 = ss * ss;
        double r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = ratpow_trunc(3.0 + s2 + r);
        t_l = r - ((t_h - 3.0) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        const double p_h = ratpow_trunc(u + v);
        const double p_l = v - (p_h - u);
        const double z_h = cp_h * p_h;
        const double z_l = cp_l * p_h + p_l * cp + dplk;
        const double t = (double)n;
        t1 = ratpow_trunc(((z_h + z_l) + dphk) + t);
        t2 = z_l - (((t1 - t) - dphk) - z_h);
    }
    /* (y1 + y2) (t1 + t2) */
    const double y1 = ratpow_trunc(y);
    double p_l = (y - y1) * t1 + y * t2;
    double p_h = y1 * t1;
    double z = p_l + p_h;
    int32_t j = ratpow_hi(z);
    const int32_t i0 = (int32_t)ratpow_lo(z);
    if (j >= 0x40900000) {                                                     /* z >= 1024 */
        if (((j - 0x40900000) | i0) != 0) return sgn * huge * huge;
        else if (p_l + ovt > z - p_h) return sgn * huge * huge;
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {                               /* z <= -1075 */
        if (((j - (int32_t)0xc090cc00) | i0) != 0) return sgn * tiny * tiny;
        else if (p_l <= z - p_h) return sgn * tiny * tiny;
    }
    /* 2^(p_h + p_l) */
    const int32_t i = j & 0x7fffffff;
    int k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {                                                      /* |z| > 0.5: n = [z + 0.5] */
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        const double t = ratpow_words(n & ~(0x000fffff >> k), 0);
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    double t = ratpow_trunc(p_l + p_h);
    const double u = t * lg2_h;
    const double v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    const double w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double r = (z * t1) / (t1 - 2.0) - (w + z * w);
    z = 1.0 - (r - z);
    j = ratpow_hi(z);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0) z = ratpow_scalbn(z, n);                               /* subnormal result */
    else z = ratpow_words(j, ratpow_lo(z));
    return sgn * z;
}
