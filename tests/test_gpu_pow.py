"""Device x^y == oracle x^y, bit for bit: both restate the reference's openlibm / fdlibm pow (csrc/rat_pow.h on the
device, oracle/fdlibm_pow.h in the checker; tests/test_cpu_pow.py pins the pair on the CPU).  Through the C ABI: the power-law family's
rollouts are pure powers and sums, so the device trajectory must equal the oracle's in every bit; its derivatives and a long
backtracking solve follow."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize("a,b", [(1.3, 1.5), (2.5, 0.7), (0.9, 1.1), (3.0, 2.0), (1.7, 0.5)])
def test_power_law_rollout_bits_equal_the_oracle(a, b):
    rng = np.random.default_rng(int(a * 100 + b * 10))
    for n in (1, 2, 4):
        N = 12
        prob = rat.PowerLawRiskSensitiveProblem(n, N, 0.01 * np.eye(n), a=a, b=b)
        ctx = rat.Context(prob)
        P = orc.Problem(prob)
        for _ in range(8):
            x0, u = rng.uniform(0.05, 1.2, n), rng.uniform(0.01, 0.9, (N, n))
            rc, xo = orc.simulate_open(P, x0, u)
            assert rc == 0
            xg = ctx.rollout_open(x0, u)
            assert np.array_equal(bits(xg), bits(xo)), (a, b, n, np.abs(xg - xo).max())


def test_power_law_feedback_rollout_and_cost_bits():
    rng = np.random.default_rng(5)
    n, N = 2, 10
    prob = rat.PowerLawRiskSensitiveProblem(n, N, 0.01 * np.eye(n))
    ctx, P = rat.Context(prob), orc.Problem(prob)
    x0, u = rng.uniform(0.2, 0.9, n), rng.uniform(0.05, 0.5, (N, n))
    _, xbar = orc.simulate_open(P, x0, u)
    L = 0.05 * rng.standard_normal((N, n, n))
    xstart = xbar.copy()
    xstart[0] += 1e-3                                  # (the feedback rollout starts from xbar[0]: perturb it so that L (x - xbar) is live)
    xg, ug = ctx.rollout_feedback(xstart, u, L)
    rc, xo, uo = orc.simulate_feedback(P, xstart, u, L)
    assert rc == 0
    # the feedback law's small matrix-vector product runs on the matrix pipe (another summation order): states agree to rounding, and
    # wherever the controls are bit-equal the next state is bit-equal (pure powers)
    assert np.abs(xg - xo).max() <= 1e-14 * max(1.0, np.abs(xo).max())
    same_u = np.all(bits(ug) == bits(uo), axis=1) & np.all(bits(xg[:-1]) == bits(xo[:-1]), axis=1)
    assert np.array_equal(bits(xg[1:][same_u]), bits(xo[1:][same_u]))
    assert ctx.integrate_cost(xo, uo) == pytest.approx(orc.integrate_cost(P, xo, uo)[1], rel=1e-14)


def test_long_backtracking_power_law_solve_matches_in_every_count():
    """The class of VERDICT r03's power-law mismatches: many iterations with heavy backtracking -- iteration and line-search counts and the
    eps sequence equal the oracle's, the value to 1e-11 (was 2e-9 .. 5e-8 with the device library's pow)."""
    rng = np.random.default_rng(77)
    worst = 0.0
    for trial in range(6):
        n, N = 2, 10
        prob = rat.PowerLawRiskSensitiveProblem(n, N, 0.01 * np.eye(n), a=float(rng.uniform(1.1, 1.6)), b=float(rng.uniform(1.2, 1.8)),
                                                p=float(rng.uniform(2.1, 2.9)))
        P = orc.Problem(prob)
        x0, u = rng.uniform(0.0, 0.6, n), rng.uniform(0.05, 0.4, (N, n))
        theta = np.array([0.0, 0.2, 0.45])
        vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta)
        ctx = rat.Context(prob, max_batch=3)
        vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
        assert np.array_equal(so, sg) and np.array_equal(io, ig) and np.array_equal(lo, lg), (trial, so, sg, io, ig, lo, lg)
        fin = np.isfinite(vo)
        if fin.any():
            worst = max(worst, float(np.abs(vg[fin] - vo[fin]).max() / np.abs(vo[fin]).max()))
    assert worst < 1e-11, worst
