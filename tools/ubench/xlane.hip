// cross-lane primitives for the LDS-free row exchange of the symmetric sweep: (a) broadcast of a 16-lane row to all four rows with
// v_permlane32_swap + v_permlane16_swap (gfx950), (b) broadcast of one lane of each row along the row with DPP row_newbcast.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void rows01_23(double x, double &e, double &o, int kg) {
    // returns e = row kg of x on every row, o = row kg+1 of x on every row (kg = 0 or 2)
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    // step 1: A = [lo32 | lo32], B = [hi32 | hi32] (halves of the wave)
    auto s32l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto s32h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    unsigned al = kg == 0 ? s32l[0] : s32l[1], ah = kg == 0 ? s32h[0] : s32h[1];     // rows (kg, kg+1, kg, kg+1)
    auto s16l = __builtin_amdgcn_permlane16_swap(al, al, false, false);
    auto s16h = __builtin_amdgcn_permlane16_swap(ah, ah, false, false);
    e = __hiloint2double(s16h[0], s16l[0]);
    o = __hiloint2double(s16h[1], s16l[1]);
}
template <int K> __device__ __forceinline__ double colbc(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double *out) {
    const int l = threadIdx.x;
    double x = 100.0 * (l >> 4) + (l & 15);
    double e0, o0, e2, o2;
    rows01_23(x, e0, o0, 0);
    rows01_23(x, e2, o2, 2);
    out[l] = e0; out[64 + l] = o0; out[128 + l] = e2; out[192 + l] = o2;
    out[256 + l] = colbc<5>(x); out[320 + l] = colbc<12>(x);
}
int main() {
    double *d, h[384]; (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int g = l >> 4, j = l & 15;
        bad += h[l] != 0.0 + j; bad += h[64 + l] != 100.0 + j; bad += h[128 + l] != 200.0 + j; bad += h[192 + l] != 300.0 + j;
        bad += h[256 + l] != 100.0 * g + 5; bad += h[320 + l] != 100.0 * g + 12;
    }
    printf("mismatches: %d\n", bad);
    for (int q = 0; q < 6; ++q) { printf("out%d:", q); for (int l = 0; l < 64; l += 7) printf(" %g", h[64 * q + l]); printf("\n"); }
    return 0;
}
