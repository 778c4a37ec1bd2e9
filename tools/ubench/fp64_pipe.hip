// Do the f64 matrix pipe (v_mfma_f64_16x16x4) and the f64 vector ALU overlap inside one SIMD, or do they share the datapath?
// One wave (and then 2 / 4 waves on one SIMD) runs: (a) independent MFMAs only, (b) independent v_fma_f64 only, (c) both interleaved.
// (c) ~ max(a, b): separate pipes; (c) ~ a + b: one FP64 datapath.  hipcc --offload-arch=gfx950 -O3 fp64_pipe.hip -o fp64_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define NV 16      // v_fma_f64 per MFMA in the mixed loop: 16 x 4 cycles = 64 cycles = one MFMA

template <int MODE>   // 0 mfma, 1 valu f64, 2 mfma + valu f64, 3 valu f32, 4 mfma + valu f32
__global__ void k(double *out, unsigned long long *cyc, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double v[NV];
    float w[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = threadIdx.x * 1e-9 + i; w[i] = threadIdx.x * 1e-3f + i; }
    const double c = 1.0000001;
    const float cf = 1.0001f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 0 || MODE == 2 || MODE == 4) { if (q == 0) a0 = MFMA(a, b, a0); if (q == 1) a1 = MFMA(a, b, a1); if (q == 2) a2 = MFMA(a, b, a2); if (q == 3) a3 = MFMA(a, b, a3); }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = fma(v[i], c, 1e-9);
            }
            if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int i = 0; i < NV; ++i) w[i] = fmaf(w[i], cf, 1e-3f);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i] + w[i];
    asm volatile("" :: "v"(a0[0]), "v"(a1[0]), "v"(a2[0]), "v"(a3[0]), "v"(s));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0[0] + a1[1] + a2[2] + a3[3] + s;
    if ((threadIdx.x & 63) == 0) { cyc[threadIdx.x >> 6] = t1 - t0; cyc[32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); }
}
int main() {
    double *out; unsigned long long *cyc, h[64];
    hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 64);
    const int iters = 2000;
    const char *names[5] = {"mfma f64 only (4 per iteration)", "v_fma_f64 only (64 per iteration)", "mfma f64 + v_fma_f64 interleaved", "v_fma_f32 only (64 per iteration)", "mfma f64 + v_fma_f32 interleaved"};
    for (int waves = 1; waves <= 16; waves *= 2) {
        printf("%d wave(s) in one workgroup:\n", waves);
        for (int mode = 0; mode < 5; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 8 * 64, hipMemcpyDeviceToHost);
            printf("  %-36s cycles per iteration / simd id, per wave:", names[mode]);
            for (int w = 0; w < waves; ++w) printf(" %.0f/%d", (double)h[w] / iters, (int)((h[32 + w] >> 4) & 3));
            printf("\n");
        }
    }
    return 0;
}
