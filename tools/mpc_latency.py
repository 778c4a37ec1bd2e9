"""Latency of ONE solve on a fresh (x_0, u_array) -- what a receding-horizon caller pays every control step: rat_ileqg_solve_batch with B = 1 on
a new x_0 each call, with the first batch's initialize! rolled out inside the solve kernel (switch init_lazy = 1, default) and by the shared
rollout launch in front of it (0).    python tools/mpc_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat

prob, x0, u = rat.synthetic_lq_problem()
rng = np.random.default_rng(3)
xs = [x0 + 0.05 * rng.standard_normal(x0.size) for _ in range(200)]
for lazy in (1, 0):
    ctx = rat.Context(prob, max_batch=1)
    ctx.debug_set("init_lazy", lazy)
    th = np.array([1.5])
    for x in xs[:20]:
        ctx.solve_batch(x, u, th)
    t0 = time.perf_counter()
    for x in xs[20:]:
        v = ctx.solve_batch(x, u, th)
    dt = (time.perf_counter() - t0) / (len(xs) - 20)
    print(f"init_lazy = {lazy}: {dt * 1e3:.4f} ms per solve on a fresh x_0 (host call, arrays in and out); value {v[0][0]:.12f}")
