import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ratilqr.jl_amd as rat
from oracle import oracle as orc
from test_gpu_wide import wide_problem, theta_grid
for (n, m, N, seed, kappa, tv) in [(13, 4, 20, 1, 0.0, False), (16, 4, 30, 2, 0.0, False), (14, 2, 25, 3, 0.02, False), (15, 3, 20, 4, 0.0, True), (16, 1, 12, 5, 0.03, True)]:
    prob, x0, u = wide_problem(n, m, N, seed, kappa, tv)
    P = orc.Problem(prob)
    theta = theta_grid(P, x0, u)
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta, nthreads=8)
    res = {}
    for sw in (0, 1):
        os.environ["RATILQR_WIDE16"] = str(sw)
        ctx = rat.Context(prob, max_batch=theta.size)
        del os.environ["RATILQR_WIDE16"]
        vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
        fin = np.isfinite(vo)
        err = np.abs(vg[fin] / vo[fin] - 1).max() if np.array_equal(np.isfinite(vg), fin) else np.nan
        print(f"n {n} m {m} N {N} tv {tv} kappa {kappa} wide16 {sw}: status_eq {np.array_equal(sg, so)} iters_eq {np.array_equal(ig, io)} ls_eq {np.array_equal(lg, lo)} max rel err {err:.2e}", flush=True)
        if not np.array_equal(sg, so): print("   ", sg, so)
# timing
for N in (50, 25):
    prob, x0, u = wide_problem(16, 4, N, 7)
    B = 1024
    theta = 0.2 * np.abs(1 + 2 * np.random.default_rng(0).standard_normal(B))
    for sw in (0, 1):
        os.environ["RATILQR_WIDE16"] = str(sw)
        ctx = rat.Context(prob, max_batch=B); del os.environ["RATILQR_WIDE16"]
        ctx.solve_batch(x0, u, theta)
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(3): out = ctx.solve_batch(x0, u, theta)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        km = ctx.profile_get()["solve_wide"]["ms"] / 3
        print(f"16x4 N={N} B=1024 wide16 {sw}: {ms:.3f} ms per batch (kernel {km:.3f} ms), {B / ms:.1f} k solves/s; iters mean {out[2].mean():.2f} ls {out[3].mean():.2f}")
