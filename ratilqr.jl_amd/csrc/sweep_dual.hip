// sweep_dual.hip -- kernel wrapper and launcher of the dual-recursion sweep (body: sweep_dual.h, also a phase of solve_fused_kernel)
#include "sweep_dual.h"

template <bool WTV, bool HASL>
__global__ __launch_bounds__(64) void sweep_dual_kernel(SweepArgs a) {
    __shared__ double wls[WLS_DUAL];
    sweep_dual_body<WTV, HASL>(a, blockIdx.x, wls);
}

void launch_sweep_dual(const SweepArgs &a, int nsamples, hipStream_t s) {
    if (nsamples <= 0) return;
    const dim3 grid(nsamples), block(64);
    if (a.mode == 7) {
        if (a.pb.W_tv) hipLaunchKernelGGL((sweep_dual_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((sweep_dual_kernel<false, true>), grid, block, 0, s, a);
    } else {
        if (a.pb.W_tv) hipLaunchKernelGGL((sweep_dual_kernel<true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((sweep_dual_kernel<false, false>), grid, block, 0, s, a);
    }
}
