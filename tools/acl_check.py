"""A/B of the block kernel's closed-loop rollouts: deviation form (block_acl = 1) against the round-2/3 split rollouts (0) and the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc
for kappa in (0.0, 0.05):
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
    for B in (128, 512):
        th = np.abs(1 + 2 * np.random.default_rng(1).standard_normal(B)) + 0.01
        res = {}
        for acl in (1, 0):
            ctx = rat.Context(prob, max_batch=B)
            ctx.debug_set("block_acl", acl)
            res[acl] = ctx.solve_batch(x0, u, th)
            t0 = time.perf_counter()
            for _ in range(200):
                ctx.solve_batch(x0, u, th)
            print(f"kappa {kappa} B {B} acl {acl}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per host-buffer batch, path {ctx.get_path(B)}")
        (v1, s1, i1, l1), (v0, s0, i0, l0) = res[1], res[0]
        fin = np.isfinite(v0)
        print("   acl vs split: max rel diff", np.abs(v1[fin] - v0[fin]).max() / np.abs(v0[fin]).max(), "| statuses / iterations / line-search counts equal:",
              np.array_equal(s1, s0), np.array_equal(i1, i0), np.array_equal(l1, l0))
        if B == 128:
            vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, th, nthreads=8)
            print("   acl vs oracle: max rel diff", np.abs(v1[fin] - vo[fin]).max() / np.abs(vo[fin]).max(), np.array_equal(s1, so), np.array_equal(i1, io), np.array_equal(l1, lo))
