// HBM streaming-write ceiling for the rollin kernels' tile records: hipMemsetAsync vs 8 B/lane and 16 B/lane store kernels vs the rollin
// pattern.  NTRAJ=1024 (default, 172 MB: inside the 256 MB MALL) / 8192 (1.4 GB: what a round of the E = 8 path writes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void w8(double *p, long n) { long i = blockIdx.x * (long)blockDim.x + threadIdx.x; long s = (long)gridDim.x * blockDim.x; for (; i < n; i += s) p[i] = 1.0; }
__global__ void w16(double2 *p, long n) { long i = blockIdx.x * (long)blockDim.x + threadIdx.x; long s = (long)gridDim.x * blockDim.x; for (; i < n; i += s) p[i] = make_double2(1.0, 2.0); }
// the rollin pattern: 1024 waves, each writing its own 168 KB region in 417-double steps of 8 B/lane stores
__global__ void wroll(double *p) { double *q = p + (long)blockIdx.x * 21007; int l = threadIdx.x; for (int t = 0; t < 50; ++t) { double *tp = q + t * 417;
    tp[l] = 1; tp[64 + l] = 1; tp[128 + l] = 1; tp[192 + l] = 1; tp[256 + l] = 1; if (l < 16) tp[320 + l] = 1; tp[336 + l] = 1; if (l < 16) tp[400 + l] = 1; if (l == 0) tp[416] = 1; } }
int main() {
    const long n = (getenv("NTRAJ") ? atol(getenv("NTRAJ")) : 1024L) * 21007; double *p; hipMalloc(&p, n * 8 + 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
#define T(name, stmt) for (int r = 0; r < 3; ++r) { hipEventRecord(a); stmt; hipEventRecord(b); hipEventSynchronize(b); } hipEventElapsedTime(&ms, a, b); printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, n * 8 / (ms * 1e-3) / 1e12);
    T("hipMemsetAsync", hipMemsetAsync(p, 0, n * 8))
    T("8 B/lane grid-stride", hipLaunchKernelGGL(w8, dim3(2048), dim3(256), 0, 0, p, n))
    T("16 B/lane grid-stride", hipLaunchKernelGGL(w16, dim3(2048), dim3(256), 0, 0, (double2 *)p, n / 2))
    T("rollin pattern (1024 waves)", hipLaunchKernelGGL(wroll, dim3((unsigned)(n / 21007)), dim3(64), 0, 0, p))
    return 0;
}
