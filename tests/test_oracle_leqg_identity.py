"""theta > 0 pins of the ORACLE that do not depend on anyone's reading of the Julia.

The reference's own known-answer tests pin the recursion hard at theta = 0 (K5: gains == LQR Riccati) but at theta > 0 only through
K7 (theta = 1e-8 ~ theta = 0), K11 (a sign) and K13 (self-consistency): a transposed D or a wrong factor on theta/2 s'M^-1 s would pass
all of them.  Here the value of the returned affine policy is computed from first principles (tests/leqg_exact.py: the total cost as ONE
quadratic form in the stacked noise, closed-form Gaussian integral) and compared with the oracle's sweeps and solves; the gain formula
L = -H^-1 G, dl = -H^-1 g with D S (ileqg.jl:367-382) is pinned by stationarity of that exact value in dl_t and L_t at the sweep's output.
The GPU twin is tests/test_gpu_leqg_identity.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from leqg_exact import breakdown_theta, exact_value, random_lq

SHAPES = [(4, 2, 20), (6, 3, 15), (12, 4, 50)]          # (n, m, N); the last one is BASELINE config 2's size
RTOL = 1e-10


def _setup(n, m, N, seed):
    prob, x0, u = random_lq(n, m, N, seed)
    P = orc.Problem(prob)
    rc, xbar = orc.simulate_open(P, x0, u)
    assert rc == 0
    rc, ap = orc.approximate_model(P, u, xbar)
    assert rc == 0
    th_bd = breakdown_theta(prob, x0, u, np.zeros((N, m, n)), xbar)       # of the open-loop policy (what initialize! evaluates)
    return prob, P, x0, u, xbar, ap, th_bd


@pytest.mark.parametrize("shape", SHAPES)
def test_sweep_values_are_the_exact_risk_sensitive_values(shape):
    n, m, N = shape
    prob, P, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=100 + n)
    assert 0.05 < th_bd < 1e4
    for frac in (0.0, 0.2, 0.5, 0.8, 0.95):
        theta = frac * th_bd
        # initialize!'s open-loop policy evaluation (L = 0, dl = nothing)   ileqg.jl:234
        rc, d0 = orc.dp_eval(P, ap, np.zeros((N, m, n)), None, theta, 0.0)
        ex0, ok = exact_value(prob, x0, u, None, np.zeros((N, m, n)), xbar, theta)
        assert rc == 0 and ok and abs(d0["s"][0] - ex0) <= RTOL * abs(ex0), (frac, d0["s"][0], ex0)
        # gain sweep (mu = 0: no regularisation) and the policy evaluation of its own output   ileqg.jl:341-406 / 412-465
        rc, L, dl, dg, mu, _ = orc.dp_gain(P, ap, theta, mu=0.0)
        assert rc == 0 and mu == 0.0
        ex, ok = exact_value(prob, x0, u, dl, L, xbar, theta)
        assert ok and abs(dg["s"][0] - ex) <= RTOL * abs(ex), (frac, dg["s"][0], ex)
        rc, de = orc.dp_eval(P, ap, L, dl, theta, 0.0)
        assert rc == 0 and abs(de["s"][0] - ex) <= RTOL * abs(ex)
        assert ex <= ex0 + 1e-9 * abs(ex0)                     # the optimised policy is no worse than the open-loop one


@pytest.mark.parametrize("shape", SHAPES)
def test_solve_returns_the_exact_value_of_its_policy(shape):
    n, m, N = shape
    prob, P, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=200 + n)
    for frac in (0.0, 0.3, 0.7, 0.9):
        theta = frac * th_bd
        s = orc.ILEQGSolver(P)
        rc = s.solve(x0, u, theta)
        assert rc == 0
        ex, ok = exact_value(prob, x0, s.l_array, None, s.L_array, s.x_array, theta)
        assert ok and abs(s.s.value_current - ex) <= RTOL * abs(ex), (frac, s.s.value_current, ex)


def _grad(prob, x0, u, dl, L, xbar, theta, picks_dl, picks_L, h=1e-4):
    out = []
    for (t, i) in picks_dl:
        dp, dm = dl.copy(), dl.copy()
        dp[t, i] += h
        dm[t, i] -= h
        out.append((exact_value(prob, x0, u, dp, L, xbar, theta)[0] - exact_value(prob, x0, u, dm, L, xbar, theta)[0]) / (2 * h))
    for (t, i, j) in picks_L:
        Lp, Lm = L.copy(), L.copy()
        Lp[t, i, j] += h
        Lm[t, i, j] -= h
        out.append((exact_value(prob, x0, u, dl, Lp, xbar, theta)[0] - exact_value(prob, x0, u, dl, Lm, xbar, theta)[0]) / (2 * h))
    return np.array(out)


@pytest.mark.parametrize("shape", SHAPES[:2] + [(12, 4, 20)])
def test_gains_of_the_gain_sweep_are_stationary_points_of_the_exact_value(shape):
    """For LQ problems the gain sweep's (dl, L) is the LEQG-optimal affine policy: the exact value has zero gradient there.  This pins
    L = -H^-1 G and dl = -H^-1 g WITH D S at theta > 0 (a theta = 0 gain used at theta > 0 is visibly not stationary)."""
    n, m, N = shape
    prob, P, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=300 + n)
    theta = 0.6 * th_bd
    rc, L, dl, dg, _, _ = orc.dp_gain(P, ap, theta, mu=0.0)
    assert rc == 0
    rng = np.random.default_rng(5)
    picks_dl = [(int(rng.integers(N)), int(rng.integers(m))) for _ in range(4)] + [(0, 0), (N - 1, m - 1)]
    picks_L = [(int(rng.integers(N)), int(rng.integers(m)), int(rng.integers(n))) for _ in range(4)] + [(0, 0, 0), (N - 1, m - 1, n - 1)]
    g = _grad(prob, x0, u, dl, L, xbar, theta, picks_dl, picks_L)
    scale = abs(dg["s"][0])
    assert np.max(np.abs(g)) <= 1e-6 * max(1.0, scale), g
    # teeth: the risk-neutral gains are NOT stationary for the risk-sensitive value
    rc, L0, dl0, _, _, _ = orc.dp_gain(P, ap, 0.0, mu=0.0)
    g0 = _grad(prob, x0, u, dl0, L0, xbar, theta, picks_dl, picks_L)
    assert np.max(np.abs(g0)) > 1e3 * max(np.max(np.abs(g)), 1e-12)


def test_infeasible_theta_is_infeasible_for_both():
    """Above the breakdown of the open-loop policy the Gaussian integral diverges and the oracle's initialize! must refuse (M not PD)."""
    n, m, N = 6, 3, 15
    prob, P, x0, u, xbar, ap, th_bd = _setup(n, m, N, seed=77)
    rc, _ = orc.dp_eval(P, ap, np.zeros((N, m, n)), None, 1.05 * th_bd, 0.0)
    assert rc != 0
    rc, d = orc.dp_eval(P, ap, np.zeros((N, m, n)), None, 0.98 * th_bd, 0.0)
    ex, ok = exact_value(prob, x0, u, None, np.zeros((N, m, n)), xbar, 0.98 * th_bd)
    assert rc == 0 and ok and abs(d["s"][0] - ex) <= 1e-8 * abs(ex)
