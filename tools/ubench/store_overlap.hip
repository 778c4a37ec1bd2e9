// Does a single-wave-per-SIMD kernel overlap its tile stores with a dependent ALU chain?  (rollin_kernel shape:
// 1024 waves x 50 steps; per step a ~1800-cycle dependent chain, then the 417-double tile.)
// Variants: chain only | chain + 9 x 8 B/lane stores | chain + 4 x 16 B/lane stores | both with a prefetched load
// consumed one step later (gfx9 has ONE vmcnt for loads and stores, so a load wait also drains older stores).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ST, int LD>
__global__ __launch_bounds__(64) void k(double *p, const double *src, double *out, int nf) {
    const int l = threadIdx.x;
    double *q = p + (long)blockIdx.x * 21058;
    const double *s = src + (long)blockIdx.x * 50 * 64;
    double x = 1.0 + l * 1e-9, nx = LD ? s[l] : 0.0;
    for (int t = 0; t < 50; ++t) {
        double cx = nx;
        if (LD) nx = s[(t < 49 ? t + 1 : t) * 64 + l];
        x += cx;
        for (int i = 0; i < nf; ++i) x = fma(x, 0.999999, 1e-7);
        double *tp = q + t * 418;
        if (ST == 1) {
            tp[l] = x; tp[64 + l] = x; tp[128 + l] = x; tp[192 + l] = x; tp[256 + l] = x;
            if (l < 16) tp[320 + l] = x;
            tp[336 + l] = x;
            if (l < 16) tp[400 + l] = x;
            if (l == 0) tp[416] = x;
        } else if (ST == 2) {
            double2 v = make_double2(x, x);
            double2 *t2 = (double2 *)tp;
            t2[l] = v; t2[64 + l] = v; t2[128 + l] = v;
            if (l < 17) t2[192 + l] = v;
        }
    }
    if (x == 123.456) out[0] = x;
}
int main() {
    double *p, *src, *out;
    hipMalloc(&p, 1024L * 21058 * 8 + 64); hipMalloc(&src, 1024L * 50 * 64 * 8); hipMalloc(&out, 8);
    hipMemset(src, 0, 1024L * 50 * 64 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
#define T(name, ST, LD, nf) for (int r = 0; r < 3; ++r) { hipEventRecord(a); hipLaunchKernelGGL((k<ST, LD>), dim3(1024), dim3(64), 0, 0, p, src, out, nf); hipEventRecord(b); hipEventSynchronize(b); } \
    hipEventElapsedTime(&ms, a, b); printf("%-44s nf=%3d %7.1f us\n", name, nf, ms * 1e3);
    for (int nf : {0, 70, 140}) {
        T("chain only", 0, 0, nf)
        T("chain + 9 x 8B stores", 1, 0, nf)
        T("chain + 4 x 16B stores", 2, 0, nf)
        T("chain + prefetched load", 0, 1, nf)
        T("chain + prefetched load + 9 x 8B stores", 1, 1, nf)
        T("chain + prefetched load + 4 x 16B stores", 2, 1, nf)
    }
    return 0;
}
