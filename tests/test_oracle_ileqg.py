"""Pins the CPU oracle with the reference's own known-answer tests for src/ileqg.jl.

Each block restates one @test group of /root/reference/test/ileqg_test.jl (line numbers cited);
K-numbers refer to SURVEY.md section 8c.  CPU only.
"""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc

N = 10


def lin_problem(cost="k"):
    """f = x + u, W = I, N = 10 (ileqg_test.jl:12-18); cost variants of :13-14, :53-54, :68-69."""
    I2 = np.eye(2)
    if cost == "k":            # c(k,x,u) = k ; h = 1
        return rat.LQRiskSensitiveProblem(I2, I2, Q=np.zeros((N, 2, 2)), R=np.zeros((2, 2)), N=N, W=I2,
                                          q0=np.arange(N, dtype=float), q0f=1.0)
    if cost == "xu":           # c = 0.5x'x + u'u + x'u ; h = 0.5x'x
        return rat.LQRiskSensitiveProblem(I2, I2, Q=I2, R=2 * I2, P=I2, N=N, W=I2, Qf=I2)
    if cost == "quad":         # c = 0.5x'x + u'u ; h = 0.5x'x
        return rat.LQRiskSensitiveProblem(I2, I2, Q=I2, R=2 * I2, N=N, W=I2, Qf=I2)
    raise ValueError(cost)


@pytest.fixture(scope="module")
def base():
    prob = lin_problem("k")
    P = orc.Problem(prob)
    u = np.ones((N, 2))
    rc, x = orc.simulate_open(P, np.zeros(2), u)
    assert rc == 0
    return prob, P, u, x


def test_K1_rollouts(base):                                   # ileqg_test.jl:20-29
    prob, P, u, x = base
    assert np.all(x[0] == 0)
    for t in range(N):
        assert np.all(x[t + 1] == prob.f(x[t], u[t]))
    rc, xn, un = orc.simulate_feedback(P, x, u, np.ones((N, 2, 2)))
    assert rc == 0 and np.all(un == u) and np.all(xn == x)


def test_K2_integrate_cost(base):                             # :32-33
    prob, P, u, x = base
    rc, cost = orc.integrate_cost(P, x, u)
    assert rc == 0
    assert np.isclose(cost, sum(prob.c(k, x[k], u[k]) for k in range(N)) + prob.h(x[-1]), rtol=1e-14)
    assert cost == 46.0


def test_K3_initialize(base):                                 # :36-49
    prob, P, u, x = base
    s = orc.ILEQGSolver(P)
    assert s.initialize(np.zeros(2), u, 0.0) == 0
    assert np.all(s.l_array == u) and np.all(s.L_array == 0) and np.all(s.x_array == x)
    assert s.s.mu == 0.0 and s.s.delta == s.s.o.delta_0
    assert s.s.d_current == np.inf and s.s.iter_current == 0 and s.s.n_hist == 0
    rc, ap = orc.approximate_model(P, u, x)
    rc2, dp = orc.dp_eval(P, ap, np.zeros((N, 2, 2)), None, 0.0, 0.0)
    assert rc == 0 and rc2 == 0
    assert np.isclose(s.s.value_current, dp["s"][0], rtol=1e-15)


def test_K4_approximate_model(base):                          # :53-66
    _, _, u, x = base
    P = orc.Problem(lin_problem("xu"))
    rc, ap = orc.approximate_model(P, u, x)
    a = ap.arrays()
    assert rc == 0
    ii = np.arange(1, N + 1)
    assert np.allclose(a["q"][:N], 0.5 * (2 * (ii - 1) ** 2) + 2.0 + 2 * (ii - 1), rtol=1e-14)
    assert np.isclose(a["q"][N], 0.5 * x[-1] @ x[-1])
    assert np.allclose(a["qv"][:N], x[:N] + 1.0) and np.allclose(a["qv"][N], x[N])
    assert np.allclose(a["Q"], np.eye(2)) and np.allclose(a["r"], x[:N] + 2.0)
    assert np.allclose(a["R"], 2 * np.eye(2)) and np.allclose(a["P"], np.eye(2))
    assert np.all(a["W"] == np.eye(2))
    assert np.allclose(a["A"], np.eye(2)) and np.allclose(a["B"], np.eye(2))


@pytest.fixture(scope="module")
def quad(base):
    _, _, u, x = base
    P = orc.Problem(lin_problem("quad"))
    rc, ap = orc.approximate_model(P, u, x)
    assert rc == 0
    return P, ap, u, x


def _shape_checks(dp, L, dl):                                 # :72-84 / :112-123
    assert dp["s"].shape == (N + 1,) and dp["sv"].shape == (N + 1, 2) and dp["S"].shape == (N + 1, 2, 2)
    for S in dp["S"]:
        assert np.all(S == S.T) and np.all(np.linalg.eigvalsh(S) > 0)
    assert dp["g"].shape == (N, 2) and dp["G"].shape == (N, 2, 2) and dp["H"].shape == (N, 2, 2)


def test_K5_K6_gain_sweep_matches_lqr(quad):                  # :70-108
    P, ap, u, x = quad
    rc, L, dl, dp, mu, de = orc.dp_gain(P, ap, 0.0)
    assert rc == 0 and mu == 0.0
    _shape_checks(dp, L, dl)
    a = ap.arrays()
    S = [None] * (N + 1)
    S[N] = a["Q"][N]
    for t in reversed(range(N)):                              # :89-97 (independent LQR Riccati)
        Q, R, A, B = a["Q"][t], a["R"][t], a["A"][t], a["B"][t]
        S[t] = Q + A.T @ S[t + 1] @ A - A.T @ S[t + 1] @ B @ np.linalg.solve(R + B.T @ S[t + 1] @ B, B.T @ S[t + 1] @ A)
    for t in range(N):                                        # :98-104
        R, A, B = a["R"][t], a["A"][t], a["B"][t]
        Llqr = -np.linalg.solve(R + B.T @ S[t + 1] @ B, B @ S[t + 1] @ A)
        assert np.allclose(Llqr, L[t], rtol=1e-8, atol=0)
    for t in range(N):                                        # :108  u + dl - L x == 0
        assert np.linalg.norm(u[t] + dl[t] - L[t] @ x[t]) <= 1e-8
    # SURVEY App. C anchors (independent NumPy restatement)
    assert np.isclose(dp["s"][0], 18.544703353520532, rtol=1e-13)
    assert np.isclose(L[0][0, 0], -0.49999928474460364, rtol=1e-13) and np.isclose(L[9][0, 0], -1 / 3, rtol=1e-14)
    assert np.allclose(dl[0], [-1, -1], rtol=1e-12)


def test_K7_small_theta_matches_risk_neutral(quad):           # :110-125
    P, ap, u, x = quad
    _, L0, dl0, dp0, _, _ = orc.dp_gain(P, ap, 0.0)
    rc, L2, dl2, dp2, _, _ = orc.dp_gain(P, ap, 1e-8)
    assert rc == 0
    _shape_checks(dp2, L2, dl2)
    assert np.isclose(dp0["s"][0], dp2["s"][0], rtol=1e-5)
    rt = np.sqrt(np.finfo(float).eps)
    for t in range(N):
        assert np.linalg.norm(dl0[t] - dl2[t]) <= rt * max(np.linalg.norm(dl0[t]), np.linalg.norm(dl2[t]))
    assert np.isclose(dp2["s"][0], 18.544703638236378, rtol=1e-12)
    _, L3, _, dp3, _, _ = orc.dp_gain(P, ap, 0.05)
    assert np.isclose(dp3["s"][0], 20.120819300590032, rtol=1e-13)
    assert np.isclose(L3[0][0, 0], -0.5361635533314971, rtol=1e-13)


def test_K8_policy_eval_reproduces_gain_sweep_exactly(quad):  # :127-130  (== within one implementation)
    P, ap, u, x = quad
    _, L, dl, dp, _, _ = orc.dp_gain(P, ap, 0.0)
    rc, dp3 = orc.dp_eval(P, ap, L, dl, 0.0, 0.0)
    assert rc == 0 and np.all(dp3["s"] == dp["s"])


def test_K9_line_search_linear_system(quad):                  # :133-134
    P, ap, u, x = quad
    s = orc.ILEQGSolver(P)
    assert s.initialize(np.zeros(2), u, 0.0) == 0
    assert s.s.value_current == 460.0
    _, L, dl, dp, _, _ = orc.dp_gain(P, ap, 0.0)
    s.set_L(L)
    assert s.line_search(dl, 0.0) == 0
    assert np.isclose(s.s.value_current, dp["s"][0], rtol=1.5e-8)


def test_K10_mu_delta_arithmetic(quad):                       # :137-148
    P, _, u, _ = quad
    s = orc.ILEQGSolver(P)
    s.initialize(np.zeros(2), u, 0.0)
    s.increase_mu_delta()
    assert s.s.delta == 4.0 and s.s.mu == 1e-6
    s.increase_mu_delta()
    assert s.s.delta == 8.0 and s.s.mu == 8e-6
    s = orc.ILEQGSolver(P)
    s.initialize(np.zeros(2), u, 0.0)
    s.decrease_mu_delta()
    assert s.s.delta == 0.5 and s.s.mu == 0.0


@pytest.fixture(scope="module")
def nonlinear():                                              # :151-161
    prob = rat.PowerLawRiskSensitiveProblem(2, N, 0.01 * np.eye(2), a=1.3, b=1.5, p=2.5, hconst=1.0)
    return orc.Problem(prob), 0.1 * np.ones((N, 2))


def test_K11_nonlinear_first_line_search(nonlinear):          # :163-170
    P, u = nonlinear
    s = orc.ILEQGSolver(P)
    assert s.initialize(np.zeros(2), u, 0.5) == 0
    assert s.step(0.5) == 0          # approximate_model + solve_approximate_dp! + line_search!
    h = s.eps_history
    assert len(h) == 1 and h[0, 0] == 1.0 and h[0, 1] < 0.0
    assert np.isclose(h[0, 1], -0.12464762861727641, rtol=1e-9)


def test_K12_nonlinear_solve_drives_state_to_zero(nonlinear):  # :172-174
    P, u = nonlinear
    s = orc.ILEQGSolver(P)
    assert s.solve(np.zeros(2), u, 0.0) == 0
    assert np.all(np.abs(s.x_array) <= 1e-4)
    assert s.s.iter_current == 4 and s.s.n_ls_evals == 4
    assert np.isclose(s.s.value_current, 1.0029075497782471, rtol=1e-12)


def test_solver_option_asserts(nonlinear):                    # ileqg.jl:195-201
    P, _ = nonlinear
    for bad in (dict(lam=1.0), dict(d=0.0), dict(mu_min=0.0), dict(delta_0=0.0), dict(eps_init=1.5),
                dict(eps_init=1e-7), dict(eps_min=1.0)):
        with pytest.raises(AssertionError):
            orc.ILEQGSolver(P, **bad)


def test_infeasible_theta_is_reported_at_init():              # F6: open-loop sweep decides feasibility
    prob, x0, u = rat.synthetic_lq_problem(seed=0)
    P = orc.Problem(prob)
    v, st, it, ls = orc.compute_value_batch(P, x0, u, [0.0, 1.0, 8.0, 50.0])
    assert np.all(np.isfinite(v[:3])) and np.all(st[:3] == 0) and np.all(it[:3] == 2) and np.all(ls[:3] == 2)
    assert np.isinf(v[3]) and st[3] == orc.ERR_M_NOT_PD_INIT
    assert v[0] < v[1] < v[2]


def test_noisy_rollouts_of_the_oracle(base):                  # simulate_dynamics(..., rng)  ileqg.jl:44-55, :94-109
    """The reference has no test of its rng rollouts; what can be pinned without Julia: with f = x + u and W = I the realised
    noise is the injected draw itself, zero noise reproduces the deterministic methods, the feedback law is the stated one and
    the cost of a rollout is integrate_cost of that rollout."""
    prob, P, u, x = base
    z = np.random.default_rng(5).standard_normal((9, N, 2))
    rc, xn, un, cost = orc.simulate_noisy(P, np.zeros(2), u, None, z)
    assert rc == 0 and np.all(un == u[None])
    assert np.allclose(xn[:, 1:] - xn[:, :-1] - u[None], z, rtol=0, atol=1e-13)      # chol(I) = I: w_k = z_k
    rc, x0n, _, c0 = orc.simulate_noisy(P, np.zeros(2), u, None, np.zeros((2, N, 2)))
    assert rc == 0 and np.all(x0n[0] == x) and c0[0] == 46.0                          # == K1 / K2
    Wd = np.array([[2.0, 0.6], [0.6, 1.0]])
    prob2 = rat.LQRiskSensitiveProblem(np.eye(2), np.eye(2), Q=np.eye(2), R=2 * np.eye(2), P=np.eye(2), N=N, W=Wd, Qf=np.eye(2))
    P2 = orc.Problem(prob2)
    L = 0.3 * np.ones((N, 2, 2))
    rc, xd = orc.simulate_open(P2, np.array([0.5, -1.0]), u)
    rc, xf, uf, cf = orc.simulate_noisy(P2, xd, u, L, z)
    assert rc == 0
    Lc = np.linalg.cholesky(Wd)
    for k in (0, 8):
        for t in range(N):
            assert np.allclose(uf[k, t], u[t] + L[t] @ (xf[k, t] - xd[t]), rtol=1e-14, atol=1e-14)
            assert np.allclose(xf[k, t + 1], xf[k, t] + uf[k, t] + Lc @ z[k, t], rtol=1e-13, atol=1e-13)
        rc, ck = orc.integrate_cost(P2, xf[k], uf[k])
        assert ck == cf[k]


def test_closure_path_of_the_oracle_equals_its_family_path():
    """oracle.closure_solve (the reference's solve! loop over Python closures, sweeps by orc_dp_gain / orc_dp_eval: the checker of the
    generic-closure GPU path) reproduces the C oracle's own solve on a family problem written as closures -- bit for bit."""
    rng = np.random.default_rng(2)
    n, m, N = 4, 2, 12
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    prob = rat.LQRiskSensitiveProblem(0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), Q=np.eye(n), R=0.3 * np.eye(m),
                                      P=0.05 * rng.standard_normal((m, n)), qv=0.1 * rng.standard_normal(n), rv=0.1 * rng.standard_normal(m),
                                      q0=0.2, N=N, W=1e-3 * np.eye(n), Qf=np.eye(n), kappa=0.02)
    x0, u = 0.5 * rng.standard_normal(n), np.zeros((N, m))
    A, B, kap, Q, R, P, qv, rv, q0, Qf = prob.A, prob.B, prob.kappa, prob.Q, prob.R, prob.P, prob.qv, prob.rv, float(prob.q0), prob.Qf
    cp = orc.ClosureProblem(lambda x, uu: A @ x + B @ uu + kap * x ** 3,
                            lambda k, x, uu: 0.5 * x @ Q @ x + 0.5 * uu @ R @ uu + uu @ P @ x + qv @ x + rv @ uu + q0, lambda x: 0.5 * x @ Qf @ x,
                            prob.W, N, n, m, lambda x, uu: (A + np.diag(3 * kap * x ** 2), B),
                            lambda k, x, uu: (Q @ x + P.T @ uu + qv, Q, R @ uu + P @ x + rv, R, P), lambda x: (Qf @ x, Qf))
    P_ = orc.Problem(prob)
    for theta in (0.0, 1.5, 8.0, 400.0):
        r = orc.closure_solve(cp, x0, u, theta)
        s = orc.ILEQGSolver(P_)
        rc = s.solve(x0, u, theta)
        assert r["status"] == rc
        if rc == 0:
            assert r["iters"] == s.s.iter_current and r["ls_evals"] == s.s.n_ls_evals
            assert abs(r["value"] - s.s.value_current) <= 1e-12 * abs(r["value"])
            assert np.abs(r["L"] - s.L_array).max() <= 1e-12 and np.abs(r["x"] - s.x_array).max() <= 1e-12
            assert np.array_equal(np.array(r["eps_history"])[:, 0], s.eps_history[:, 0])
        else:
            assert np.isinf(r["value"])
