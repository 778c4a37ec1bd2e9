// What does a hand-off between TWO WORKGROUPS of one cooperative launch cost on gfx950?  (a sample on two compute units --
// the evaluation team on one, the gain-sweep team on the other -- would hand trajectories / gains / control words through L2 / HBM.)
// S pairs of 256-thread workgroups ping-pong K times: the sender writes a payload of `bytes` to global memory, releases (agent scope),
// raises a word; the receiver spins on the word (relaxed agent-scope loads + s_sleep), acquires, reads the payload and checks it, answers
// with a payload of its own.  Reported: cycles (s_memtime, 100 MHz -> ns) per ONE-WAY hand-off, for partners in the same XCD
// (blockIdx b and b + S, S a multiple of 8: workgroups are dealt round-robin to the 8 XCDs) and in different XCDs (2b, 2b + 1),
// and the launch cost of hipLaunchCooperativeKernel against an ordinary launch of the same kernel.
//   hipcc --offload-arch=gfx950 -O3 xwg_handoff.hip -o xwg_handoff && ./xwg_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void spin_ge(int *w, int want) {
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) __builtin_amdgcn_s_sleep(1);
}

// mode 0: partner = b +- S (same XCD); mode 1: partner = b ^ 1 (neighbouring XCDs)
// fence 0: agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1: what the memory model prescribes between workgroups);
// fence 1: partners share an XCD, hence an L2: the writer only waits for its (write-through) stores to reach L2, the reader only drops
//          its CU's vector L1 (buffer_inv sc0) -- correct ONLY for partners in one XCD;  fence 2: flags only (payload not checked)
// fence 3: release = vmcnt(0) (write-through stores have reached the shared L2), acquire = buffer_inv sc1 (drops L1, walks L2 for non-coherent lines)
// fence 4: release = buffer_wbl2 sc1 + vmcnt(0), no acquire (what the write-back alone costs; payload not checked)
// fence 5: release = vmcnt(0), no invalidate: the reader's payload loads are agent-scope relaxed ATOMIC loads (sc1: they miss L1 by definition)
template <int FENCE>
__device__ __forceinline__ void rel() {
    if (FENCE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else if (FENCE == 4) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int FENCE>
__device__ __forceinline__ void acq() {
    if (FENCE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (FENCE == 1) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
    else if (FENCE == 3) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}
template <int FENCE>
__device__ __forceinline__ double ld(const double *p) {
    if (FENCE == 5) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int FENCE>
__global__ __launch_bounds__(256, 1) void pingpong(double *buf, int *flags, unsigned long long *out, unsigned *xcc_out, int S, int K, int ndbl, int mode) {
    __shared__ double lds[8192];                     // (keeps one workgroup per CU like the solve kernel's ~100 KB would)
    const int b = blockIdx.x;
    int pair, role;
    if (mode == 0) { pair = b % S; role = b / S; } else { pair = b >> 1; role = b & 1; }
    double *mine = buf + (size_t)(2 * pair + role) * ndbl, *theirs = buf + (size_t)(2 * pair + (role ^ 1)) * ndbl;
    int *fmine = flags + 32 * (2 * pair + role), *ftheirs = flags + 32 * (2 * pair + (role ^ 1));
    if (threadIdx.x == 0) xcc_out[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    lds[threadIdx.x] = 0.0;
    double bad = 0.0;
    unsigned long long t0 = 0;
    for (int k = 1; k <= K; ++k) {
        if (k == 2) t0 = __builtin_amdgcn_s_memrealtime();        // (first round trip: start-up skew)
        if (role == 0) {
            for (int i = threadIdx.x; i < ndbl; i += 256) mine[i] = (double)(k * 3 + i);
            rel<FENCE>();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(fmine, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) spin_ge(ftheirs, k);
            __syncthreads();
            acq<FENCE>();
            for (int i = threadIdx.x; i < ndbl; i += 256) bad += (FENCE == 2 || FENCE == 4) ? 0.0 : fabs(ld<FENCE>(theirs + i) - (double)(k * 5 + i));
        } else {
            if (threadIdx.x == 0) spin_ge(ftheirs, k);
            __syncthreads();
            acq<FENCE>();
            for (int i = threadIdx.x; i < ndbl; i += 256) bad += (FENCE == 2 || FENCE == 4) ? 0.0 : fabs(ld<FENCE>(theirs + i) - (double)(k * 3 + i));
            for (int i = threadIdx.x; i < ndbl; i += 256) mine[i] = (double)(k * 5 + i);
            rel<FENCE>();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(fmine, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < 256; ++i) s += lds[i];
        out[2 * b] = t1 - t0;
        out[2 * b + 1] = (s == 0.0) ? 0ull : 1ull;
    }
}

__global__ __launch_bounds__(256, 1) void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
    const int S = 128, K = 201;
    double *buf; int *flags; unsigned long long *out; unsigned *xcc;
    const int maxd = 8192;
    CK(hipMalloc(&buf, sizeof(double) * 2 * S * maxd));
    CK(hipMalloc(&flags, sizeof(int) * 32 * 2 * S));
    CK(hipMalloc(&out, sizeof(unsigned long long) * 4 * S));
    CK(hipMalloc(&xcc, sizeof(unsigned) * 2 * S));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("cooperative launch supported: %d\n", coop);
    const void *kern[6] = {(const void *)pingpong<0>, (const void *)pingpong<1>, (const void *)pingpong<2>, (const void *)pingpong<3>,
                           (const void *)pingpong<4>, (const void *)pingpong<5>};
    const char *fname[6] = {"agent-scope fences", "vmcnt(0) + buffer_inv sc0", "flags only (no payload check)", "vmcnt(0) + buffer_inv sc1",
                            "buffer_wbl2 sc1 only (no payload check)", "vmcnt(0) + atomic (sc1) payload loads"};
    for (int coopl = 1; coopl >= 0; --coopl)
    for (int fence = 0; fence < 6; ++fence)
    for (int mode = 0; mode < 2; ++mode) {
        if (coopl == 0 && (fence != 5 || mode != 0)) continue;            // (ordinary launch: the candidate configuration only)
        for (int ndbl : {0, 16, 1024, 2560, 8192}) {
            CK(hipMemsetAsync(flags, 0, sizeof(int) * 32 * 2 * S, st));
            int S_ = S, K_ = K, nd = ndbl, md = mode;
            void *args[] = {&buf, &flags, &out, &xcc, &S_, &K_, &nd, &md};
            if (coopl) CK(hipLaunchCooperativeKernel(kern[fence], dim3(2 * S), dim3(256), args, 0, st));
            else CK(hipLaunchKernel(kern[fence], dim3(2 * S), dim3(256), args, 0, st));
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> h(4 * S);
            std::vector<unsigned> hx(2 * S);
            CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 4 * S, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hx.data(), xcc, sizeof(unsigned) * 2 * S, hipMemcpyDeviceToHost));
            std::vector<double> per;
            int bad = 0, same = 0;
            for (int b = 0; b < 2 * S; ++b) { per.push_back((double)h[2 * b] * 10.0 / (2.0 * (K - 1))); bad += (int)h[2 * b + 1]; }
            for (int p = 0; p < S; ++p) same += mode == 0 ? (hx[p] == hx[p + S]) : (hx[2 * p] == hx[2 * p + 1]);
            std::sort(per.begin(), per.end());
            printf("%s | %s | pairs %s payload %6d B: one-way hand-off median %6.0f ns  min %6.0f  max %6.0f | mismatching workgroups %d | pairs in one XCD %d / %d\n",
                   coopl ? "coop" : "ordinary", fname[fence], mode == 0 ? "(b, b+S) " : "(2b, 2b+1)", ndbl * 8, per[per.size() / 2], per.front(), per.back(), bad, same, S);
        }
    }
    // launch cost: cooperative against ordinary (events around 200 back-to-back launches of an empty 256-workgroup kernel)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int c = 0; c < 2; ++c) {
        int *np = nullptr;
        void *args[] = {&np};
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 200; ++i) {
                if (c) CK(hipLaunchCooperativeKernel((const void *)empty_kernel, dim3(256), dim3(256), args, 0, st));
                else hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, np);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // a single launch, host-timed to completion
        printf("%s launch: %.2f us per launch back to back (events)\n", c ? "cooperative" : "ordinary   ", ms * 1000.0 / 200);
    }
    return 0;
}
