"""Wall time of the whole RAT iLQR solve (config 3 without speculation: 1024 CE samples, 100 elites, 5 CE iterations + the final
iLEQG solve at theta_opt) and of RAT iLQR++ through the C ABI, against 5 x the bare compute_cost batch: what the callers around the
hot path add.  python tools/ce_timing.py  (on an MI355X)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce, nelder_mead as nm

prob, x0, u = rat.synthetic_lq_problem()
z = np.random.default_rng(31).standard_normal(40000)
for E in (1, 8):
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=1024, num_elite=100, spec_eps=E)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ce.solve_(solver, prob, x0, u, z, kl_bound=0.1)
        ts.append(time.perf_counter() - t0)
    print(f"CE solve E={E}: theta_opt {out[0]:.6f}  n_solves {solver.c.n_solves}  wall ms per solve! call:", [round(1e3 * t, 3) for t in ts])
ctx = rat.Context(prob, max_batch=1024)
ctx.set_initial(x0, u)
th = torch.as_tensor(np.abs(1 + 2 * np.random.default_rng(1).standard_normal(1024)), dtype=torch.float64, device="cuda")
cost = torch.empty_like(th)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ctx.compute_cost_dev(th.data_ptr(), 1024, 0.1, cost.data_ptr())
    print("5 bare compute_cost_dev batches: ms", round(1e3 * (time.perf_counter() - t0), 3))
t0 = time.perf_counter(); r = ctx.solve(x0, u, 1.0); print("single solve! (theta = 1): ms", round(1e3 * (time.perf_counter() - t0), 3))
t0 = time.perf_counter(); r = ctx.solve(x0, u, 1.0); print("single solve! (theta = 1): ms", round(1e3 * (time.perf_counter() - t0), 3))
s = rat.NelderMeadBilevelOptimizationSolver()
for rep in range(3):
    t0 = time.perf_counter()
    out = nm.solve_(s, prob, x0, u, kl_bound=0.1)
    print(f"NM solve: theta_opt {out[0]:.6f} iterations {s.c.iter_current}: ms", round(1e3 * (time.perf_counter() - t0), 3))
