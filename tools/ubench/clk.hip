// shader clock actually delivered: s_memtime (clock64) ticks per s_memrealtime (wall_clock64, 100 MHz) tick while 1024 waves run f64 FMA/MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(double *out, int n, int mf) {
    double x = threadIdx.x; d4 acc = {0, 0, 0, 0};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) { x = fma(x, 0.999, 1e-3); if (mf) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0); }
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x < 4) { out[blockIdx.x * 2] = (double)(c1 - c0); out[blockIdx.x * 2 + 1] = (double)(w1 - w0); }
    if (x + acc[0] == 1.2345) out[100] = x;
}
int main() {
    double *d, h[8]; (void)hipMalloc(&d, 1024);
    for (int mf = 0; mf < 2; ++mf) for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, d, 20000, mf); (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("mfma=%d: clock64 ticks %.0f, wall ticks %.0f (100 MHz) -> clock64 rate %.1f MHz\n", mf, h[0], h[1], h[0] / h[1] * 100.0);
    }
    return 0;
}
