"""End-to-end latency of one RAT iLQR solve (rat_ce_solve: 5 CE iterations of 1024 samples + the final solve at theta_opt; what a
receding-horizon controller pays per control step) against the kernel time of its batches: how much of the wall time is host work
(draws, uploads, waits, sort / elite update) rather than solves.   python tools/ce_latency.py [num_samples]   (on an MI355X)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat                 # noqa: E402
from ratilqr.jl_amd import cross_entropy as ce  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    prob, x0, u = rat.synthetic_lq_problem()
    for E in (1,):
        solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=B, num_elite=max(3, B // 10), spec_eps=E) if "spec_eps" in rat.CrossEntropyBilevelOptimizationSolver.__init__.__code__.co_varnames else rat.CrossEntropyBilevelOptimizationSolver(num_samples=B, num_elite=max(3, B // 10))
        for rep in range(3):
            ce.solve_(solver, prob, x0, u, 1234 + rep, kl_bound=0.1)
        ts = []
        for rep in range(10):
            t0 = time.perf_counter()
            out = ce.solve_(solver, prob, x0, u, 99 + rep, kl_bound=0.1)
            ts.append(time.perf_counter() - t0)
        ctx = solver.context(prob)
        ctx.profile(True); ctx.profile_reset()
        ce.solve_(solver, prob, x0, u, 7, kl_bound=0.1)
        pr = {k: v for k, v in ctx.profile_get().items() if v["launches"]}
        ctx.profile(False)
        kms = sum(v["ms"] for v in pr.values())
        print(f"B {B} E {E}: rat_ce_solve wall {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f}) | kernels {kms:.3f} ms in "
              f"{sum(v['launches'] for v in pr.values())} launches {[(k, v['launches'], round(v['ms'], 3)) for k, v in pr.items()]} | theta_opt {out[0]:.4f}")
    # single solve latency
    ctx = rat.Context(prob)
    for _ in range(5):
        ctx.solve(x0, u, 1.0)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); ctx.solve(x0, u, 1.0); ts.append(time.perf_counter() - t0)
    print(f"rat_ileqg_solve (one sample, x / l / L returned): wall {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f})")


if __name__ == "__main__":
    main()
