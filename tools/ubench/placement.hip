// Where does the dispatcher put the wavefronts of small workgroups?  (solve_block_kernel: two-wave workgroups, <= 256 VGPRs, 33 KB LDS.)
// Every wave records HW_REG_HW_ID / HW_REG_XCC_ID and spins long enough for the whole grid to be co-resident; the host prints, per
// grid size, how the waves of a workgroup pair up on SIMDs and how many waves each SIMD of a CU ends up with.
//   hipcc --offload-arch=gfx950 -O3 placement.hip -o placement && ./placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void k(unsigned *out, int spin, double *sink) {
    __shared__ double lds[33000 / 8];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    const unsigned long long t0 = __builtin_readcyclecounter();
    double acc = threadIdx.x;
    lds[threadIdx.x] = acc;
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) acc = acc * 1.0000001 + lds[(threadIdx.x * 7) & 1023];
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * NW + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (acc == 12345.678) *sink = acc;
}
template <int NW> void run(int B, size_t dyn) {
    unsigned *d; double *s;
    hipMalloc(&d, sizeof(unsigned) * B * NW * 2); hipMalloc(&s, 8);
    hipFuncSetAttribute((const void *)k<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipLaunchKernelGGL(k<NW>, dim3(B), dim3(64 * NW), dyn, 0, d, 400000, s);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> h(B * NW * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;        // key (xcc, se, sh, cu) -> waves per simd
    std::map<unsigned, std::vector<int>> first_per_cu;  // wave 0 of each block per simd
    int pair[4][4] = {};
    for (int b = 0; b < B; ++b)
        for (int w = 0; w < NW; ++w) {
            const unsigned hw = h[(b * NW + w) * 2], xcc = h[(b * NW + w) * 2 + 1] & 15;
            const unsigned key = (xcc << 8) | ((hw >> 8) & 255), simd = (hw >> 4) & 3;
            if (per_cu[key].empty()) { per_cu[key].assign(4, 0); first_per_cu[key].assign(4, 0); }
            per_cu[key][simd]++;
            if (w == 0) first_per_cu[key][simd]++;
            if (w == 1) pair[(h[(b * NW) * 2] >> 4) & 3][simd]++;
        }
    std::map<std::string, int> hist, hist0;
    for (auto &kv : per_cu) { char buf[64]; snprintf(buf, 64, "%d,%d,%d,%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]); hist[buf]++; }
    for (auto &kv : first_per_cu) { char buf[64]; snprintf(buf, 64, "%d,%d,%d,%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]); hist0[buf]++; }
    printf("NW %d B %d dynLDS %zu (%s): %zu CUs used\n  waves per SIMD (simd0,1,2,3) x CUs:", NW, B, dyn, hipGetErrorString(e), per_cu.size());
    for (auto &kv : hist) printf("  [%s] x%d", kv.first.c_str(), kv.second);
    printf("\n  wave-0s per SIMD x CUs:");
    for (auto &kv : hist0) printf("  [%s] x%d", kv.first.c_str(), kv.second);
    // with only waves 0 and 1 of every workgroup kept alive (the others exit at once): live waves per SIMD x CUs
    std::map<unsigned, std::vector<int>> live;
    for (int b = 0; b < B; ++b)
        for (int w = 0; w < 2 && w < NW; ++w) {
            const unsigned hw = h[(b * NW + w) * 2], xcc = h[(b * NW + w) * 2 + 1] & 15;
            const unsigned key = (xcc << 8) | ((hw >> 8) & 255);
            if (live[key].empty()) live[key].assign(4, 0);
            live[key][(hw >> 4) & 3]++;
        }
    std::map<std::string, int> histl;
    for (auto &kv : live) { char buf[64]; snprintf(buf, 64, "%d,%d,%d,%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3]); histl[buf]++; }
    printf("\n  waves 0+1 only, per SIMD x CUs:");
    for (auto &kv : histl) printf("  [%s] x%d", kv.first.c_str(), kv.second);
    printf("\n  (simd of wave 0 -> simd of wave 1) counts:");
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) if (pair[a][c]) printf("  %d->%d x%d", a, c, pair[a][c]);
    printf("\n");
    hipFree(d); hipFree(s);
}
int main() {
    for (int B : {128, 256, 512, 1024}) run<2>(B, 0);
    run<2>(256, 49024); run<2>(512, 21973);
    for (int B : {128, 256, 512}) run<3>(B, 0);
    for (int B : {128, 256}) run<5>(B, 0);
    for (int B : {128, 256}) run<8>(B, 0);
    for (int B : {256, 512, 768, 1024}) run<4>(B, 0);
    for (int B : {512, 768}) run<3>(B, 0);
    return 0;
}
