// sweep_dual.hip -- kernel wrapper and launcher of the dual-recursion sweep (body: sweep_dual.h, also a phase of solve_fused_kernel)
#include "sweep_dual.h"

template <bool WTV>
__global__ __launch_bounds__(64) void sweep_dual_kernel(SweepArgs a) {
    sweep_dual_body<WTV>(a, blockIdx.x);
}

void launch_sweep_dual(const SweepArgs &a, int nsamples, hipStream_t s) {
    if (nsamples <= 0) return;
    if (a.pb.W_tv) hipLaunchKernelGGL((sweep_dual_kernel<true>), dim3(nsamples), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((sweep_dual_kernel<false>), dim3(nsamples), dim3(64), 0, s, a);
}
