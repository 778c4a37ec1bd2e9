// sweep_dual.hip -- kernel wrapper and launcher of the dual-recursion sweep (body: sweep_dual.h, also a phase of solve_fused_kernel)
#include "sweep_dual.h"

template <int WM, bool HASL, int FLY = 0>
__global__ __launch_bounds__(64) void sweep_dual_kernel(SweepArgs a) {
    __shared__ double wls[WLS_DUAL];
    sweep_dual_body<WM, HASL, FLY>(a, blockIdx.x, wls);
}

void launch_sweep_dual(const SweepArgs &a, int nsamples, hipStream_t s) {
    if (nsamples <= 0) return;
    const dim3 grid(nsamples), block(64);
#define DUAL_LAUNCH(W) do { \
        if (a.mode == 7 && a.fly) {                    /* candidate 0's record holds only [c_x | c_u | c]: tiles formed in the sweep */ \
            if (a.pb.cost_tv) hipLaunchKernelGGL((sweep_dual_kernel<W, true, 2>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((sweep_dual_kernel<W, true, 1>), grid, block, 0, s, a); \
        } else if (a.mode == 7) hipLaunchKernelGGL((sweep_dual_kernel<W, true>), grid, block, 0, s, a); \
        else if (a.fly) {                              /* initialize!'s trajectory came from the shared slot: [x; u] and cost-gradient rows only */ \
            if (a.pb.cost_tv) hipLaunchKernelGGL((sweep_dual_kernel<W, false, 2>), grid, block, 0, s, a); \
            else hipLaunchKernelGGL((sweep_dual_kernel<W, false, 1>), grid, block, 0, s, a); \
        } else hipLaunchKernelGGL((sweep_dual_kernel<W, false>), grid, block, 0, s, a); } while (0)
    if (a.pb.W_tv) DUAL_LAUNCH(1); else if (a.pb.W_diag) DUAL_LAUNCH(2); else DUAL_LAUNCH(0);
#undef DUAL_LAUNCH
}
