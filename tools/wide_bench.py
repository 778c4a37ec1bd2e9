"""Throughput of the general-size solve kernel (wide.hip) on one GPU: LQ-plus-noise problems of SURVEY 8(d)'s recipe at n x m beyond the
12 + 4 tile, CE batch 1024, against the C oracle on the host threads.  Not the headline (bench.py): a functionality path, measured once."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat                 # noqa: E402
from oracle import oracle as orc             # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for n, m, N in ((12, 4, 50), (16, 4, 50), (20, 6, 50), (24, 8, 50), (32, 8, 50), (32, 32, 50)):
        prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=N)
        theta = np.abs(1.0 + 2.0 * np.random.default_rng(1).standard_normal(B)) * (0.2 if n > 12 else 1.0)
        ctx = rat.Context(prob, max_batch=B)
        v, st, it, ls = ctx.solve_batch(x0, u, theta)
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.solve_batch(x0, u, theta)
            t.append(time.perf_counter() - t0)
        dt = min(t)
        nb = min(B, 64)
        t0 = time.perf_counter()
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta[:nb], nthreads=min(16, os.cpu_count()))
        tc = time.perf_counter() - t0
        ok = np.array_equal(so, st[:nb]) and np.array_equal(io, it[:nb])
        fin = np.isfinite(vo)
        err = np.abs(v[:nb][fin] / vo[fin] - 1).max() if fin.any() else 0.0
        print(f"n {n:2d} m {m:2d} N {N} B {B}: {dt * 1e3:8.2f} ms/batch = {B / dt / 1e3:8.1f} k solves/s | feasible {int((st == 0).sum())}/{B} iters {int(it.max())} | "
              f"oracle {nb / tc / 1e3:6.2f} k solves/s on {min(16, os.cpu_count())} threads | parity {ok} rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
