"""The general-size solve kernel under the profiler (tools/profile_r03_aux.sh): CE batch 1024 at 16 x 4 and 32 x 32, no oracle."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
out = []
sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(16, 4), (32, 32)]      # e.g. 16x4
for n, m in sizes:
    prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=50)
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(1).standard_normal(1024)) * 0.2
    ctx = rat.Context(prob, max_batch=1024)
    ctx.solve_batch(x0, u, theta)
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); v, st, it, ls = ctx.solve_batch(x0, u, theta); t.append(time.perf_counter() - t0)
    out.append({"n": n, "m": m, "ms_per_batch": min(t) * 1e3, "solves_per_s": 1024 / min(t), "mean_iters": float(np.mean(it))})
print(json.dumps(out))
