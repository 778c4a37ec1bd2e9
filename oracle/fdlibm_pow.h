/* fdlibm_pow.h -- TEST INFRASTRUCTURE (CPU oracle): Float64 ^ Float64 as the reference's Julia computes it.
 *
 * Third-party arithmetic on the path: `x.^1.3`, `u.^1.5`, `x.^2.5` of the reference's nonlinear test system (test/ileqg_test.jl:151-155)
 * and every ForwardDiff derivative of them (a x^(a-1), a (a-1) x^(a-2)) go through Base.:^(::Float64, ::Float64), which Julia 1.5.x
 * lowers to llvm.pow.f64 -> `pow` of the openlibm it ships (openlibm v0.7.x, src/e_pow.c = FreeBSD msun / Sun fdlibm __ieee754_pow
 * "1.5 04/04/22").  openlibm is not under /root/reference and not in this image, so its published algorithm is restated here; the
 * oracle no longer calls the host's glibc pow (a different algorithm: the two agree to <= 1 ulp and differ on ~9 % of arguments).
 * Pins: tests/test_cpu_pow.py -- <= 1 ulp from glibc on 4 x 10^6 arguments, every IEEE special case equal, the hexadecimal and
 * decimal spellings of the 22 published constants agree, and bit-identity with the product's own restatement (csrc/rat_pow.h).
 *
 * Method (fdlibm's own description):
 *   1. Compute and return log2(x) in two pieces: log2(x) = w1 + w2, where w1 has 53-24 = 29 bit trailing zeros.
 *   2. Perform y*log2(x) = n+y' by simulating multi-precision arithmetic, where |y'| <= 0.5.
 *   3. Return x**y = 2**n*exp(y'*log2).
 * Only double operations in the stated order: compiled with -ffp-contract=off (oracle/Makefile).
 *
 * fdlibm's notice, preserved as its licence asks (the algorithm, its step order and its published constants are fdlibm's):
 * ====================================================
 * Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
 *
 * Developed at SunSoft, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this
 * software is freely granted, provided that this notice
 * is preserved.
 * ====================================================
 */
#ifndef ORC_FDLIBM_POW_H
#define ORC_FDLIBM_POW_H
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef union { double d; struct { uint32_t lo, hi; } w; } orc_dw;        /* little-endian words of a double */
#define ORC_HI(x_, out_) do { orc_dw u_; u_.d = (x_); (out_) = (int32_t)u_.w.hi; } while (0)
#define ORC_LO(x_, out_) do { orc_dw u_; u_.d = (x_); (out_) = u_.w.lo; } while (0)
#define ORC_SET_HI(x_, v_) do { orc_dw u_; u_.d = (x_); u_.w.hi = (uint32_t)(v_); (x_) = u_.d; } while (0)
#define ORC_SET_LO(x_, v_) do { orc_dw u_; u_.d = (x_); u_.w.lo = (uint32_t)(v_); (x_) = u_.d; } while (0)

static double orc_scale2(double z, int n) {          /* z 2^n, subnormal result range (one rounding) */
    double f = 0.0;
    if (n < -1000) { double g = 0.0; ORC_SET_HI(g, (0x3ff - 1000) << 20); z *= g; n += 1000; }
    ORC_SET_HI(f, (uint32_t)(0x3ff + n) << 20);
    return z * f;
}

static double orc_pow(double x, double y) {
    static const double bp[2] = {1.0, 1.5};
    static const double dp_h[2] = {0.0, 5.84962487220764160156e-01}, dp_l[2] = {0.0, 1.35003920212974897128e-08};
    static const double zero = 0.0, one = 1.0, two = 2.0, two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300;
    static const double L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                        L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
                        ovt = 8.0085662595372944372e-0017,
                        cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01, cp_l = -7.02846165095275826516e-09,
                        ivln2 = 1.44269504088896338700e+00, ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
    double z, ax, z_h, z_l, p_h, p_l;
    double y1, t1, t2, r, s, t, u, v, w;
    int32_t i, j, k, yisint, n;
    int32_t hx, hy, ix, iy;
    uint32_t lx, ly;

    ORC_HI(x, hx); ORC_LO(x, lx);
    ORC_HI(y, hy); ORC_LO(y, ly);
    ix = hx & 0x7fffffff; iy = hy & 0x7fffffff;

    if ((iy | ly) == 0) return one;                          /* y == zero: x**0 = 1 */
    if (hx == 0x3ff00000 && lx == 0) return one;             /* x == 1: 1**y = 1, even if y is NaN */
    if (ix > 0x7ff00000 || ((ix == 0x7ff00000) && (lx != 0)) || iy > 0x7ff00000 || ((iy == 0x7ff00000) && (ly != 0)))
        return (x + 0.0) + (y + 0.0);                        /* y != zero: NaN if either argument is NaN */

    /* determine if y is an odd int when x < 0: yisint = 0 (not an integer), 1 (odd), 2 (even) */
    yisint = 0;
    if (hx < 0) {
        if (iy >= 0x43400000) yisint = 2;                    /* even integer y */
        else if (iy >= 0x3ff00000) {
            k = (iy >> 20) - 0x3ff;                          /* exponent */
            if (k > 20) {
                uint32_t jj = ly >> (52 - k);
                if ((jj << (52 - k)) == ly) yisint = 2 - (int32_t)(jj & 1);
            } else if (ly == 0) {
                j = iy >> (20 - k);
                if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
            }
        }
    }

    /* special value of y */
    if (ly == 0) {
        if (iy == 0x7ff00000) {                              /* y is +-inf */
            if (((ix - 0x3ff00000) | lx) == 0) return one;   /* (-1)**+-inf is 1 */
            else if (ix >= 0x3ff00000) return (hy >= 0) ? y : zero;      /* (|x|>1)**+-inf = inf,0 */
            else return (hy < 0) ? -y : zero;                /* (|x|<1)**-,+inf = inf,0 */
        }
        if (iy == 0x3ff00000) {                              /* y is +-1 */
            if (hy < 0) return one / x; else return x;
        }
        if (hy == 0x40000000) return x * x;                  /* y is 2 */
        if (hy == 0x3fe00000) {                              /* y is 0.5 */
            if (hx >= 0) return sqrt(x);                     /* x >= +0 */
        }
    }

    ax = fabs(x);
    /* special value of x */
    if (lx == 0) {
        if (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000) {
            z = ax;                                          /* x is +-0, +-inf, +-1 */
            if (hy < 0) z = one / z;                         /* z = (1/|x|) */
            if (hx < 0) {
                if (((ix - 0x3ff00000) | yisint) == 0) z = (z - z) / (z - z);       /* (-1)**non-int is NaN */
                else if (yisint == 1) z = -z;                /* (x<0)**odd = -(|x|**odd) */
            }
            return z;
        }
    }

    n = (int32_t)((uint32_t)hx >> 31) - 1;
    if ((n | yisint) == 0) return (x - x) / (x - x);         /* (x<0)**(non-int) is NaN */
    s = one;                                                 /* s (sign of result -ve**odd) = -1 else = 1 */
    if ((n | (yisint - 1)) == 0) s = -one;                   /* (-ve)**(odd int) */

    if (iy > 0x41e00000) {                                   /* |y| is huge: > 2**31 */
        if (iy > 0x43f00000) {                               /* |y| > 2**64, must o/uflow */
            if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
            if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
        }
        /* over/underflow if x is not close to one */
        if (ix < 0x3fefffff) return (hy < 0) ? s * huge * huge : s * tiny * tiny;
        if (ix > 0x3ff00000) return (hy > 0) ? s * huge * huge : s * tiny * tiny;
        /* now |1-x| is tiny <= 2**-20, suffice to compute log(x) by x-x^2/2+x^3/3-x^4/4 */
        t = ax - one;                                        /* t has 20 trailing zeros */
        w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
        u = ivln2_h * t;                                     /* ivln2_h has 21 sig. bits */
        v = t * ivln2_l - w * ivln2;
        t1 = u + v;
        ORC_SET_LO(t1, 0);
        t2 = v - (t1 - u);
    } else {
        double ss, s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00100000) { ax *= two53; n -= 53; ORC_HI(ax, ix); }     /* take care of subnormal numbers */
        n += ((ix) >> 20) - 0x3ff;
        j = ix & 0x000fffff;
        /* determine interval */
        ix = j | 0x3ff00000;                                 /* normalize ix */
        if (j <= 0x3988E) k = 0;                             /* |x| < sqrt(3/2) */
        else if (j < 0xBB67A) k = 1;                         /* |x| < sqrt(3)   */
        else { k = 0; n += 1; ix -= 0x00100000; }
        ORC_SET_HI(ax, ix);

        /* compute ss = s_h + s_l = (x-1)/(x+1) or (x-1.5)/(x+1.5) */
        u = ax - bp[k];                                      /* bp[0] = 1.0, bp[1] = 1.5 */
        v = one / (ax + bp[k]);
        ss = u * v;
        s_h = ss;
        ORC_SET_LO(s_h, 0);
        /* t_h = ax + bp[k] High */
        t_h = zero;
        ORC_SET_HI(t_h, ((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18));
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        /* compute log(ax) */
        s2 = ss * ss;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = 3.0 + s2 + r;
        ORC_SET_LO(t_h, 0);
        t_l = r - ((t_h - 3.0) - s2);
        /* u + v = ss * (1 + ...) */
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        /* 2/(3log2) * (ss + ...) */
        p_h = u + v;
        ORC_SET_LO(p_h, 0);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;                                    /* cp_h + cp_l = 2/(3*log2) */
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        /* log2(ax) = (ss + ..) * 2/(3*log2) = n + dp_h + z_h + z_l */
        t = (double)n;
        t1 = (((z_h + z_l) + dp_h[k]) + t);
        ORC_SET_LO(t1, 0);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }

    /* split up y into y1 + y2 and compute (y1 + y2) * (t1 + t2) */
    y1 = y;
    ORC_SET_LO(y1, 0);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    ORC_HI(z, j); { uint32_t il; ORC_LO(z, il); i = (int32_t)il; }
    if (j >= 0x40900000) {                                   /* z >= 1024 */
        if (((j - 0x40900000) | i) != 0) return s * huge * huge;          /* if z > 1024: overflow */
        else { if (p_l + ovt > z - p_h) return s * huge * huge; }        /* overflow */
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {             /* z <= -1075 */
        if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;  /* z < -1075: underflow */
        else { if (p_l <= z - p_h) return s * tiny * tiny; }             /* underflow */
    }
    /* compute 2**(p_h + p_l) */
    i = j & 0x7fffffff;
    k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {                                    /* if |z| > 0.5, set n = [z + 0.5] */
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;                /* new k for n */
        t = zero;
        ORC_SET_HI(t, (n & ~(0x000fffff >> k)));
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = p_l + p_h;
    ORC_SET_LO(t, 0);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - two) - (w + z * w);
    z = one - (r - z);
    ORC_HI(z, j);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0) z = orc_scale2(z, n);                /* subnormal output */
    else ORC_SET_HI(z, j);
    return s * z;
}
#endif
