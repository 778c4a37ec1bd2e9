"""PETS rollout kernels' time against the horizon (N = 2 ... 120) at 256 / 10 k / 40 k trajectories: the slope is the time per step, the intercept
the launch and the mean kernel -- tells a latency-bound launch (a lone group) from a throughput-bound one.   python tools/pets_horizon_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import pets
n, m = 12, 4
r = np.random.default_rng(8)
A = 0.9 * np.linalg.qr(r.standard_normal((n, n)))[0]
B = r.standard_normal((n, m)) / np.sqrt(n)
x0 = r.standard_normal(n)
for Nh in (2, 10, 30, 60, 120):
    prob = rat.LQGenerativeProblem(A, B, Nh, ("gaussian", np.zeros(n), 0.03 * np.eye(n)), Q=np.eye(n), R=0.1 * np.eye(m), Qf=np.eye(n), kappa=-0.01)
    for S, K in ((100, 100), (16, 16), (400, 100)):
        ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, m)), np.stack([np.eye(m)] * Nh), num_control_samples=S, num_trajectory_samples=K)
        ctx = ds.context(prob)
        ctrl = 0.3 * r.standard_normal((S, Nh, m))
        for _ in range(3):
            c = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11)
        ctx.profile(True)
        ms = []
        for i in range(20):
            ctx.profile_reset(); c = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11 + i)
            ms.append(ctx.profile_get()["pets"]["ms"])
        print(f"N={Nh} S={S} K={K} ({S*K} trajectories, {(S*K+15)//16} waves): pets kernels {np.median(ms)*1e3:.1f} us (min {np.min(ms)*1e3:.1f})", flush=True)
