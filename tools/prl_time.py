"""The time-parallel closed-loop rollout of solve_block_psw_kernel (rollprl_body, switch psw_prl) against the one-wave rollout: kernel time of
a whole batch by HIP events, counts against each other and values to 1e-12; both against the CPU oracle.    python tools/prl_time.py [B ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat
from oracle import oracle as orc


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [1, 16, 128]
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.0)
    for B in Bs:
        theta = np.abs(1.0 + 2.0 * np.random.default_rng(B).standard_normal(B))
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=16)
        res = {}
        for prl in (0, 1):
            for duo in (0, 1):
                ctx = rat.Context(prob, max_batch=B)
                ctx.debug_set("psw_prl", prl)
                ctx.debug_set("psw_duo", duo)
                ctx.set_initial(x0, u)
                for _ in range(5):
                    out = ctx.solve_batch(x0, u, theta)
                ctx.profile(True); ctx.profile_reset()
                ms = []
                for _ in range(50):
                    ctx.profile_reset()
                    out = ctx.solve_batch(x0, u, theta)
                    ms.append(ctx.profile_get()["solve_block"]["ms"])
                fin = np.isfinite(vo)
                ok = np.array_equal(out[1], so) and np.array_equal(out[2], io) and np.array_equal(out[3], lo)
                err = np.abs(out[0][fin] - vo[fin]).max() / np.abs(vo[fin]).max()
                res[(prl, duo)] = (np.median(ms), np.min(ms), ok, err)
        for duo in (0, 1):
            (m0, n0, k0, e0), (m1, n1, k1, e1) = res[(0, duo)], res[(1, duo)]
            print(f"B={B} duo={duo}: one-wave rollout {m0:.4f} ms (min {n0:.4f}, oracle counts {k0}, err {e0:.1e})   "
                  f"time-parallel {m1:.4f} ms (min {n1:.4f}, oracle counts {k1}, err {e1:.1e})  x{m0 / m1:.3f}", flush=True)


if __name__ == "__main__":
    main()
