"""Batches beyond one sample per SIMD: the two-per-SIMD fused kernel (switch fused_occ2 = -1, the default) beside the paired kernel run in
generations (fused_occ2 = 0), headline problem and its cubic-drift variant.      python tools/occ2_time.py [B ...]      (on an MI355X)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ratilqr.jl_amd as rat

Bs = [int(a) for a in sys.argv[1:]] or [1536, 2048, 4096, 8192]
for kappa, N in ((0.0, 50), (0.06, 50), (0.0, 100)):
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa, N=N)
    for B in Bs:
        th = torch.as_tensor(np.abs(1 + 2 * np.random.default_rng(1).standard_normal(B)), dtype=torch.float64, device="cuda")
        ref, row = None, []
        for occ in ("0", "-1"):
            os.environ["RATILQR_FUSED_OCC2"] = occ
            ctx = rat.Context(prob, max_batch=B)
            del os.environ["RATILQR_FUSED_OCC2"]
            ctx.set_initial(x0, u)
            cost = torch.empty(B, dtype=torch.float64, device="cuda")
            for _ in range(5):
                ctx.compute_cost_dev(th.data_ptr(), B, 0.1, cost.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.compute_cost_enqueue(th.data_ptr(), B, 0.1, cost.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            c = cost.cpu().numpy()
            same = ref is None or np.array_equal(c, ref, equal_nan=True)
            ref = c
            row.append(f"fused_occ2 = {occ:>2}: {ms:7.3f} ms {B / ms / 1e3:6.3f} M solves/s{'' if same else ' MISMATCH'}")
            del ctx
        print(f"kappa {kappa:4.2f} N {N:3d} B {B:5d} | " + " | ".join(row), flush=True)
