"""The band in which the device's isposdef(M) decision (leading minors of the 2 x 2-block elimination, csrc/device_utils.h) and the
oracle's (Cholesky-style pivots: the reference's LAPACK path, ileqg.jl:366) may disagree, as a tested BOUND (VERDICT r02 weak #1 iii):
each side's feasibility threshold theta* of initialize! is bisected to the last bit; the two must lie within 16 ulp (<= 2e-15
relative; measured 1-3 ulp, profiles/r02_isposdef_band.md) and no status of a complete solve may differ on a grid of half-width
1e-9 theta* around it -- on every execution path, and for both arithmetic forms of (D S)[A|B] (diagonal-W folding on and off)."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _threshold(feasible, lo, hi):
    assert feasible(lo) and not feasible(hi)
    while True:
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            return lo
        if feasible(mid):
            lo = mid
        else:
            hi = mid


@pytest.mark.parametrize("kw", [dict(), dict(kappa=0.05), dict(w=1e-2), dict(n=4, m=2, N=20, seed=1), dict(n=7, m=3, N=33, seed=5)])
@pytest.mark.parametrize("wdiag", ["1", "0"])
def test_feasibility_thresholds_agree_to_a_few_ulp(kw, wdiag, monkeypatch):
    monkeypatch.setenv("RATILQR_WDIAG", wdiag)
    prob, x0, u = rat.synthetic_lq_problem(**kw)
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=44)
    to = _threshold(lambda th: orc.compute_value_batch(P, x0, u, np.array([th]))[1][0] != 1, 0.0, 1e4)
    td = _threshold(lambda th: ctx.solve_batch(x0, u, np.array([th]))[1][0] != 1, 0.0, 1e4)
    assert abs(td - to) <= 16 * np.spacing(to), (to, td, abs(td - to) / np.spacing(to))
    grid = to * (1.0 + np.linspace(-1e-9, 1e-9, 41))
    grid = grid[np.abs(grid - to) > 32 * np.spacing(to)]                  # (the tie itself is excluded: that IS the band)
    so = orc.compute_value_batch(P, x0, u, grid, nthreads=8)[1]
    for path in ("block", "fused", "rounds"):
        ctx.set_path(path)
        sd = ctx.solve_batch(x0, u, grid)[1]
        assert np.array_equal(sd, so), (path, grid[sd != so], to)
    assert (so == 1).any() and (so != 1).any()
