"""Width of the band in which the device's isposdef(M) decision (leading 2x2-block minors inside the elimination, device_utils.h) and
the oracle's (Cholesky-style pivots, the reference's LAPACK path ileqg.jl:366) can disagree, measured as |theta_dev* - theta_orc*| /
theta_orc* where theta* is each side's feasibility threshold of initialize! found by bisection to the last bit.
  python tools/pd_band.py      (on an MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc


def threshold(feasible, lo, hi):
    assert feasible(lo) and not feasible(hi)
    while True:
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            return lo
        if feasible(mid):
            lo = mid
        else:
            hi = mid


rows = []
cases = [("SURVEY 8d LQ problem (n=12, m=4, N=50)", dict()), ("kappa = 0.05", dict(kappa=0.05)), ("w = 1e-2", dict(w=1e-2)),
         ("n=4, m=2, N=20", dict(n=4, m=2, N=20, seed=1)), ("n=7, m=3, N=33, seed 5", dict(n=7, m=3, N=33, seed=5)),
         ("N=100", dict(N=100, seed=2))]
for name, kw in cases:
    prob, x0, u = rat.synthetic_lq_problem(**kw)
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=4)

    def f_orc(th):
        return orc.compute_value_batch(P, x0, u, np.array([th]))[1][0] != 1

    def f_dev(th):
        return ctx.solve_batch(x0, u, np.array([th]))[1][0] != 1

    to, td = threshold(f_orc, 0.0, 1e4), threshold(f_dev, 0.0, 1e4)
    ulps = abs(td - to) / np.spacing(to)
    # full-solve statuses on a fine grid around the threshold: where do the two sides differ at all?
    grid = to * (1.0 + np.linspace(-1e-9, 1e-9, 41))
    so = orc.compute_value_batch(P, x0, u, grid, nthreads=8)[1]
    sd = np.concatenate([ctx.solve_batch(x0, u, grid[i:i + 4])[1] for i in range(0, grid.size, 4)])
    differ = grid[(so != sd)]
    band = 0.0 if differ.size == 0 else float(np.max(np.abs(differ - to)) / to)
    rows.append((name, to, td, abs(td - to) / to, ulps, band))
    print(f"{name:42s} theta*_oracle {to:.17g}  theta*_device {td:.17g}  rel.diff {abs(td - to) / to:.2e} ({ulps:.0f} ulp)  "
          f"statuses differ within +-1e-9: {'nowhere' if differ.size == 0 else f'up to {band:.1e} from theta*'}", flush=True)
print("max relative width of the disagreement band:", max(r[3] for r in rows))
