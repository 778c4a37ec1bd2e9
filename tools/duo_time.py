"""solve_block_psw_kernel with one and with two workgroups (compute units) per sample (switch psw_duo): kernel time of a whole batch by HIP
events, counts and values against each other.    python tools/duo_time.py [B ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128]
    kappa = float(os.environ.get("KAPPA", "0"))
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
    for B in Bs:
        theta = np.abs(1.0 + 2.0 * np.random.default_rng(B).standard_normal(B))
        res = {}
        for duo in (0, 1):
            ctx = rat.Context(prob, max_batch=B)
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
            res[duo] = (out, np.median(ms), np.min(ms), ctx.debug_get("psw_duo_count"))
        (o0, m0, n0, c0), (o1, m1, n1, c1) = res[0], res[1]
        same = all(np.array_equal(a, b) for a, b in zip(o0[1:], o1[1:]))
        fin = np.isfinite(o0[0])
        err = np.abs(o0[0][fin] - o1[0][fin]).max() / np.abs(o0[0][fin]).max()
        print(f"B={B}: one workgroup {m0:.4f} ms (min {n0:.4f})  two {m1:.4f} ms (min {n1:.4f})  x{m0 / m1:.2f}  pairs formed {c1} of {55 * B}  same_counts {same}  value_err {err:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
