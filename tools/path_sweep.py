"""Execution paths of a CE batch side by side: solve_fused_kernel (one wave per sample, paired recursions), solve_block_kernel (one
workgroup per sample: a wave per candidate + a gain wave) and the round-based path, over batch sizes and speculation widths, on the
headline LQ workload and on a workload whose line search backtracks.  Results are identical on every path (tests/test_gpu_block.py).
  python tools/path_sweep.py [--batches 128 256 512 1024 2048 4096] [--widths 1 2 4 8] [--reps 30]      (on an MI355X)"""
import argparse
import os
os.environ.setdefault("RATILQR_SPEC_FORCE", "1")     # handles of width E > 1 run the speculative kernels here (spec_eps is otherwise an upper bound)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ratilqr.jl_amd as rat

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="*", default=[128, 256, 512, 1024, 2048, 4096])
ap.add_argument("--widths", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--kappas", type=float, nargs="*", default=[0.0, 0.06])
a = ap.parse_args()

PATHS = {"fused": {"RATILQR_BLOCK": "0"}, "block": {"RATILQR_BLOCK": "1"}, "rounds": {"RATILQR_FUSED": "0"}}
if os.environ.get("SWEEP_OCC2") == "1":      # the 256-register one-recursion-per-pass fused variant (two samples per SIMD) beside the default fused kernel
    PATHS = {"fused": {"RATILQR_BLOCK": "0", "RATILQR_FUSED_OCC2": "0"}, "occ2": {"RATILQR_BLOCK": "0", "RATILQR_FUSED_OCC2": "1"},
             "nodual": {"RATILQR_BLOCK": "0", "RATILQR_FUSED_DUAL": "0"}}


def theta_for(B, kappa):
    if kappa == 0.0:
        rng = np.random.default_rng(1000)
        out = []
        while len(out) < B:
            z = 1.0 + 2.0 * rng.standard_normal(B)
            out.extend(z[z > 0.0].tolist())
        return np.array(out[:B])
    return np.where(np.arange(B) % 2 == 0, 2.0, 5.0).astype(float)       # 10 / 13 line-search evaluations over 5 iterations


for kappa in a.kappas:
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
    for E in a.widths:
        for B in a.batches:
            th = torch.as_tensor(theta_for(B, kappa), dtype=torch.float64, device="cuda")
            row, ref = [], None
            for name, env in PATHS.items():
                if name in ("fused", "occ2", "nodual") and E != 1:
                    continue
                for k, v in env.items():
                    os.environ[k] = v
                ctx = rat.Context(prob, max_batch=B, spec_eps=E)
                for k in env:
                    del os.environ[k]
                ctx.set_initial(x0, u)
                cost = torch.empty(B, dtype=torch.float64, device="cuda")
                t_c = time.perf_counter()
                while time.perf_counter() - t_c < 0.15:
                    ctx.compute_cost_dev(th.data_ptr(), B, 0.1, cost.data_ptr())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    ctx.compute_cost_enqueue(th.data_ptr(), B, 0.1, cost.data_ptr())
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / a.reps * 1e3
                c = cost.cpu().numpy()
                same = ref is None or np.array_equal(c, ref, equal_nan=True)
                ref = c if ref is None else ref
                row.append(f"{name} {ms:7.3f} ms {B / ms * 1e3 / 1e6:6.3f} M/s{'' if same else ' MISMATCH'}")
                del ctx
            print(f"kappa {kappa:4.2f} E {E} B {B:5d} | " + " | ".join(row), flush=True)
