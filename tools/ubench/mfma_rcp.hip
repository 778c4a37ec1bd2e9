// Microbenchmarks that size the sweep kernel: f64 MFMA issue interval vs dependent latency, v_rcp_f64 accuracy,
// LDS exchange round trip inside one wave.  hipcc --offload-arch=gfx950 -O3 mfma_rcp.hip -o mfma_rcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__global__ void k_dep(double *out, unsigned long long *cyc, int iters) {
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { acc = MFMA(a, b, acc); acc = MFMA(a, b, acc); acc = MFMA(a, b, acc); acc = MFMA(a, b, acc); }
    asm volatile("" :: "v"(acc[0]));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep(double *out, unsigned long long *cyc, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { a0 = MFMA(a, b, a0); a1 = MFMA(a, b, a1); a2 = MFMA(a, b, a2); a3 = MFMA(a, b, a3); }
    asm volatile("" :: "v"(a0[0]), "v"(a1[0]), "v"(a2[0]), "v"(a3[0]));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent through a VALU op on the result (the real pattern: product -> scale -> next product)
__global__ void k_dep_valu(double *out, unsigned long long *cyc, int iters) {
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { acc = MFMA(a, b, acc); b = acc[0] * 0.5; acc = MFMA(a, b, acc); b = acc[1] * 0.5; acc = MFMA(a, b, acc); b = acc[2] * 0.5; acc = MFMA(a, b, acc); b = acc[3] * 0.5; }
    asm volatile("" :: "v"(acc[0]));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc[0] + b;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp(const double *x, double *r0, double *r1, double *r2) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    double p = x[i];
    double y = __builtin_amdgcn_rcp(p);
    r0[i] = y;
    double e = fma(-p, y, 1.0); y = fma(y, e, y); r1[i] = y;
    e = fma(-p, y, 1.0); y = fma(y, e, y); r2[i] = y;
}
__global__ void k_lds(double *out, unsigned long long *cyc, int iters) {
    __shared__ double buf[2][64];
    double v = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        buf[i & 1][threadIdx.x] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        v = buf[i & 1][(threadIdx.x + 17) & 63] + 1.0;
    }
    asm volatile("" :: "v"(v));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_chain(double *out, unsigned long long *cyc, int iters) {
    double v = threadIdx.x * 1e-9, c = 1.0000001;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { v = fma(v, c, 1e-9); v = fma(v, c, 1e-9); v = fma(v, c, 1e-9); v = fma(v, c, 1e-9); }
    asm volatile("" :: "v"(v));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rcp_chain(double *out, unsigned long long *cyc, int iters) {
    double v = 1.5 + threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { v = __builtin_amdgcn_rcp(v); v = __builtin_amdgcn_rcp(v); v = __builtin_amdgcn_rcp(v); v = __builtin_amdgcn_rcp(v); }
    asm volatile("" :: "v"(v));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_readlane_chain(double *out, unsigned long long *cyc, int iters) {
    double v = 1.5 + threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), 5), hi = __builtin_amdgcn_readlane(__double2hiint(v), 5);
        v = fma(v, __hiloint2double(hi, lo), 1e-9);
    }
    asm volatile("" :: "v"(v));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 65536 * 8); hipMalloc(&cyc, 8);
    const int it = 1000;
#define RUN(k, name, per) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, it); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.1f cycles per op\n", name, (double)h / (it * per));
    RUN(k_dep, "mfma f64 dependent chain", 4)
    RUN(k_indep, "mfma f64 4 independent", 4)
    RUN(k_dep_valu, "mfma f64 -> valu -> mfma", 4)
    RUN(k_lds, "lds write->read roundtrip", 1)
    RUN(k_fma_chain, "v_fma_f64 dependent", 4)
    RUN(k_rcp_chain, "v_rcp_f64 dependent", 4)
    RUN(k_readlane_chain, "readlane x2 + fma chain", 1)
    // rcp accuracy
    const int n = 65536;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX - 0.5) * 60.0);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / hx[i];
        e0 = fmax(e0, (double)fabsl((h0[i] - ex) / ex)); e1 = fmax(e1, (double)fabsl((h1[i] - ex) / ex)); e2 = fmax(e2, (double)fabsl((h2[i] - ex) / ex));
    }
    printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
