"""GPU twin of tests/test_oracle_leqg_identity.py: the HIP path (through the C ABI) against the first-principles Gaussian integral of
tests/leqg_exact.py -- not against the oracle -- at theta > 0, with time-varying cost tables, P != 0, linear terms and time-varying
non-diagonal W(k).  Pins every theta > 0 term of the device recursion and, through stationarity, the device gain formula."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from leqg_exact import breakdown_theta, exact_value, random_lq

pytestmark = pytest.mark.gpu
SHAPES = [(4, 2, 20), (6, 3, 15), (12, 4, 50)]
RTOL = 1e-10


def _setup(n, m, N, seed):
    prob, x0, u = random_lq(n, m, N, seed)
    ctx = rat.Context(prob, max_batch=8)
    xbar = ctx.rollout_open(x0, u)
    ap = ctx.approximate_model(u, xbar)
    th_bd = breakdown_theta(prob, x0, u, np.zeros((N, m, n)), xbar)
    return prob, ctx, x0, u, xbar, ap, th_bd


@pytest.mark.parametrize("shape", SHAPES)
def test_device_sweep_values_are_the_exact_risk_sensitive_values(shape):
    n, m, N = shape
    prob, ctx, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=100 + n)
    for frac in (0.0, 0.2, 0.5, 0.8, 0.95):
        theta = frac * th_bd
        st, d0 = ctx.dp_policy_eval(ap, np.zeros((N, m, n)), None, theta, 0.0)            # initialize!'s sweep (ileqg.jl:234)
        ex0, ok = exact_value(prob, x0, u, None, np.zeros((N, m, n)), xbar, theta)
        assert st == 0 and ok and abs(d0.s_array[0] - ex0) <= RTOL * abs(ex0), (frac, d0.s_array[0], ex0)
        st, L, dl, dg, mu, _ = ctx.dp_gain_sweep(ap, theta, 0.0, 2.0)                      # solve_approximate_dp! (:341-406)
        ex, ok = exact_value(prob, x0, u, dl, L, xbar, theta)
        assert st == 0 and mu == 0.0 and ok and abs(dg.s_array[0] - ex) <= RTOL * abs(ex), (frac, dg.s_array[0], ex)
        st, de = ctx.dp_policy_eval(ap, L, dl, theta, 0.0)                                 # solve_approximate_dp (:412-465)
        assert st == 0 and abs(de.s_array[0] - ex) <= RTOL * abs(ex)


@pytest.mark.parametrize("shape", SHAPES)
def test_device_solve_returns_the_exact_value_of_its_policy(shape):
    """rat_ileqg_solve (the fused single-launch solve) and the batch entry point: `value` is the exact risk-sensitive value of the
    returned affine policy (l_array, L_array) around x_array."""
    n, m, N = shape
    prob, ctx, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=200 + n)
    thetas = np.array([0.0, 0.3, 0.7, 0.9]) * th_bd
    vb, stb, _, _ = ctx.solve_batch(x0, u, thetas)
    for i, theta in enumerate(thetas):
        r = ctx.solve(x0, u, theta)
        assert r["status"] == 0 and stb[i] == 0 and r["value"] == vb[i]
        ex, ok = exact_value(prob, x0, r["l"], None, r["L"], r["x"], theta)
        assert ok and abs(r["value"] - ex) <= RTOL * abs(ex), (theta, r["value"], ex)


@pytest.mark.parametrize("shape", SHAPES[:2] + [(12, 4, 20)])
def test_device_gains_are_stationary_points_of_the_exact_value(shape):
    n, m, N = shape
    prob, ctx, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=300 + n)
    theta = 0.6 * th_bd
    st, L, dl, dg, _, _ = ctx.dp_gain_sweep(ap, theta, 0.0, 2.0)
    assert st == 0
    rng = np.random.default_rng(5)
    h = 1e-4
    worst = 0.0
    for _ in range(6):
        t, i, j = int(rng.integers(N)), int(rng.integers(m)), int(rng.integers(n))
        dp, dm = dl.copy(), dl.copy()
        dp[t, i] += h
        dm[t, i] -= h
        worst = max(worst, abs(exact_value(prob, x0, u, dp, L, xbar, theta)[0] - exact_value(prob, x0, u, dm, L, xbar, theta)[0]) / (2 * h))
        Lp, Lm = L.copy(), L.copy()
        Lp[t, i, j] += h
        Lm[t, i, j] -= h
        worst = max(worst, abs(exact_value(prob, x0, u, dl, Lp, xbar, theta)[0] - exact_value(prob, x0, u, dl, Lm, xbar, theta)[0]) / (2 * h))
    assert worst <= 1e-6 * max(1.0, abs(dg.s_array[0])), worst


def test_time_varying_linear_cost_terms_reach_the_device_in_time_major_order():
    """Regression (found by the identity above): (N, n) tables of q_vec / r_vec are Vector{Vector}s -- time slowest -- not matrices."""
    n, m, N = 4, 2, 20
    prob, ctx, x0, u, xbar, ap, _ = _setup(n, m, N, seed=9)
    want = np.array([prob.c(k, xbar[k], u[k]) for k in range(N)] + [prob.h(xbar[N])])
    assert np.max(np.abs(ap.q_array - want)) <= 1e-12 * np.max(np.abs(want))
    assert abs(ctx.integrate_cost(xbar, u) - want.sum()) <= 1e-12 * abs(want.sum())
