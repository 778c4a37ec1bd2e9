// Issue cost of prefetched tile loads in a single-wave-per-SIMD kernel (sweep_kernel shape: 1024 waves x 50 steps,
// a long dependent chain per step, tile of the NEXT step prefetched at the top of the step).
// Variants: no loads | 9 x 8 B/lane loads | 5 x 16 B/lane loads (same bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LD>
__global__ __launch_bounds__(64) void k(const double *src, double *out, int nf) {
    const int l = threadIdx.x;
    const double *s = src + (long)blockIdx.x * 50 * 576;
    double x = 1.0 + l * 1e-9;
    double n[10]; for (int i = 0; i < 10; ++i) n[i] = 0.0;
    for (int t = 49; t >= 0; --t) {
        double c[10]; for (int i = 0; i < 10; ++i) c[i] = n[i];
        const double *tp = s + (long)(t > 0 ? t - 1 : 0) * 576;
        if (LD == 1) { for (int i = 0; i < 9; ++i) n[i] = tp[64 * i + l]; }
        if (LD == 2) { const double2 *t2 = (const double2 *)tp; for (int i = 0; i < 4; ++i) { double2 v = t2[64 * i + l]; n[2 * i] = v.x; n[2 * i + 1] = v.y; } n[8] = tp[512 + l]; }
        for (int i = 0; i < 10; ++i) x += c[i];
#pragma unroll 10
        for (int i = 0; i < nf; ++i) x = fma(x, 0.999999, 1e-7);
    }
    if (x == 123.456) out[0] = x;
}
int main() {
    double *src, *out;
    (void)hipMalloc(&src, 1024L * 50 * 576 * 8); (void)hipMalloc(&out, 8);
    (void)hipMemset(src, 0, 1024L * 50 * 576 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float ms;
#define T(name, LD, nf) for (int r = 0; r < 3; ++r) { (void)hipEventRecord(a); hipLaunchKernelGGL((k<LD>), dim3(1024), dim3(64), 0, 0, src, out, nf); (void)hipEventRecord(b); (void)hipEventSynchronize(b); } \
    (void)hipEventElapsedTime(&ms, a, b); printf("%-28s nf=%3d %7.1f us\n", name, nf, ms * 1e3);
    for (int nf : {0, 100, 200}) {
        T("chain only", 0, nf)
        T("chain + 9 x 8B loads", 1, nf)
        T("chain + 4 x 16B + 1 x 8B loads", 2, nf)
    }
    return 0;
}
