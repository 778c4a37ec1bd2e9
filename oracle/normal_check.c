/* normal_check.c -- TEST INFRASTRUCTURE (tests/test_cpu_normal.py): the device generators' Box-Muller transform (csrc/rat_normal.h, plain C)
 * evaluated on the host, so that it can be held against libm.  Not part of the product; nothing here is used by the library. */
#include "../ratilqr.jl_amd/csrc/rat_normal.h"

void orc_normal_parts(const double *u1, const double *u2, long n, double *lg, double *rt, double *sn, double *cs, double *z0, double *z1) {
    for (long i = 0; i < n; ++i) {
        lg[i] = ratn_log(1.0 - u1[i]);
        rt[i] = ratn_sqrt(-2.0 * lg[i]);
        ratn_sincospi(2.0 * u2[i], &sn[i], &cs[i]);
        ratn_box_muller(u1[i], u2[i], &z0[i], &z1[i]);
    }
}
