"""Generic-closure problems: host rollouts + linearisation from the user's closures, Riccati sweeps on
the device through the operator ABI.  Reference behaviour: ileqg.jl:24-31, :71-79, :302-311 (f_returns_jacobian), :265-273 (AD)."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def lq_closures(prob):
    """The LQ + cubic-drift family written as plain closures with exact derivatives."""
    A, B, kap = prob.A, prob.B, prob.kappa
    Q, R, P, qv, rv, q0, Qf = prob.Q, prob.R, prob.P, prob.qv, prob.rv, float(prob.q0), prob.Qf        # (constant cost tables)

    def f(x, u, f_returns_jacobian=False):
        xn = A @ x + B @ u + kap * x ** 3
        return (xn, A + np.diag(3 * kap * x ** 2), B) if f_returns_jacobian else xn

    def c(k, x, u):
        return 0.5 * x @ Q @ x + 0.5 * u @ R @ u + u @ P @ x + qv @ x + rv @ u + q0

    def cd(k, x, u):
        return Q @ x + P.T @ u + qv, Q, R @ u + P @ x + rv, R, P

    def h(x):
        return 0.5 * x @ Qf @ x

    def hd(x):
        return Qf @ x, Qf

    return f, c, cd, h, hd


def small_lq():
    rng = np.random.default_rng(2)
    n, m, N = 4, 2, 12
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    prob = rat.LQRiskSensitiveProblem(0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), Q=np.eye(n), R=0.3 * np.eye(m),
                                      P=0.05 * rng.standard_normal((m, n)), qv=0.1 * rng.standard_normal(n), rv=0.1 * rng.standard_normal(m),
                                      q0=0.2, N=N, W=1e-3 * np.eye(n), Qf=np.eye(n), kappa=0.02)
    return prob, 0.5 * rng.standard_normal(n), np.zeros((N, m))


@pytest.mark.parametrize("theta", [0.0, 1.5])
def test_generic_closures_with_exact_derivatives_reproduce_the_device_family(theta):
    prob, x0, u = small_lq()
    f, c, cd, h, hd = lq_closures(prob)
    gen = rat.GenericRiskSensitiveProblem(f, c, h, prob.W, prob.N, prob.n, prob.m, f_returns_jacobian=True,
                                          c_derivatives=cd, h_derivatives=hd)
    s_dev, s_gen = rat.ILEQGSolver(prob), rat.ILEQGSolver(gen, f_returns_jacobian=True)
    x1, l1, L1, v1, h1 = rat.solve_(s_dev, prob, x0, u, theta=theta)
    x2, l2, L2, v2, h2 = rat.solve_(s_gen, gen, x0, u, theta=theta)
    assert s_dev.iter_current == s_gen.iter_current and [a[0] for a in h1] == [a[0] for a in h2]
    assert abs(v1 - v2) <= 1e-10 * abs(v1) and rel(x2, x1) < 1e-10 and rel(l2, l1) < 1e-9 and rel(L2, L1) < 1e-9
    # the stateless reference functions dispatch on the problem type
    xs = rat.simulate_dynamics(gen, x0, u)
    assert rel(xs, rat.simulate_dynamics(prob, x0, u)) < 1e-13
    assert abs(rat.integrate_cost(gen, xs, u) - rat.integrate_cost(prob, xs, u)) < 1e-12
    ap_g, ap_d = rat.approximate_model(gen, u, xs), rat.approximate_model(prob, u, xs)
    for name in ("q_array", "q_vec_array", "Q_array", "r_array", "R_array", "P_array", "A_array", "B_array", "W_array"):
        assert rel(getattr(ap_g, name), getattr(ap_d, name)) < 1e-12, name


def test_generic_closures_with_finite_differences():
    """No derivatives supplied at all: central differences stand in for the reference's ForwardDiff."""
    prob, x0, u = small_lq()
    f, c, cd, h, hd = lq_closures(prob)
    gen = rat.GenericRiskSensitiveProblem(lambda x, uu: f(x, uu), c, h, prob.W, prob.N, prob.n, prob.m)
    xs = rat.simulate_dynamics(prob, x0, u)
    ap_g, ap_d = rat.approximate_model(gen, u, xs), rat.approximate_model(prob, u, xs)
    for name in ("q_vec_array", "Q_array", "r_array", "R_array", "P_array", "A_array", "B_array"):
        assert np.allclose(getattr(ap_g, name), getattr(ap_d, name), rtol=0, atol=2e-5), name
    s_dev, s_gen = rat.ILEQGSolver(prob), rat.ILEQGSolver(gen)
    _, _, _, v1, _ = rat.solve_(s_dev, prob, x0, u, theta=1.0)
    _, _, _, v2, _ = rat.solve_(s_gen, gen, x0, u, theta=1.0)
    assert abs(v1 - v2) <= 1e-5 * abs(v1)


def test_generic_nonlinear_closure_outside_the_families():
    """A pendulum-like system no compiled-in family covers: the solve lowers the risk-sensitive value and the returned policy
    is the stated affine feedback law."""
    n, m, N, dt = 2, 1, 25, 0.1

    def f(x, u, f_returns_jacobian=False):
        xn = np.array([x[0] + dt * x[1], x[1] + dt * (-np.sin(x[0]) - 0.1 * x[1] + u[0])])
        if not f_returns_jacobian:
            return xn
        return xn, np.array([[1.0, dt], [-dt * np.cos(x[0]), 1.0 - 0.1 * dt]]), np.array([[0.0], [dt]])

    c = lambda k, x, u: 0.5 * (x @ x) + 0.05 * (u @ u)
    cd = lambda k, x, u: (x, np.eye(2), 0.1 * u, 0.1 * np.eye(1), np.zeros((1, 2)))
    h = lambda x: 2.0 * (x @ x)
    hd = lambda x: (4.0 * x, 4.0 * np.eye(2))
    gen = rat.GenericRiskSensitiveProblem(f, c, h, lambda k: 1e-3 * np.eye(2), N, n, m, f_returns_jacobian=True, c_derivatives=cd, h_derivatives=hd)
    x0, u0 = np.array([1.0, 0.0]), np.zeros((N, m))
    s = rat.ILEQGSolver(gen, f_returns_jacobian=True)
    rat.initialize_ileqg_(s, gen, x0, u0, 0.5)
    v_init = s.value_current
    x, l, L, v, hist = rat.solve_(s, gen, x0, u0, theta=0.5)
    assert v < v_init and s.iter_current >= 2 and all(e[1] < 0 or abs(e[1]) < 1e-6 * abs(v) for e in hist[:1])
    xn, un = rat.simulate_dynamics(gen, x, l, L)
    assert rel(xn, x) < 1e-12 and rel(un, l) < 1e-12          # the nominal trajectory is a fixed point of its own policy


# ---- against the ORACLE's closure path (not HIP against HIP), and the batch entry points ---------------------------
from oracle import oracle as orc  # noqa: E402


def pendulum():
    n, m, N, dt = 2, 1, 25, 0.1

    def f(x, u, f_returns_jacobian=False):
        xn = np.array([x[0] + dt * x[1], x[1] + dt * (-np.sin(x[0]) - 0.1 * x[1] + u[0])])
        if not f_returns_jacobian:
            return xn
        return xn, np.array([[1.0, dt], [-dt * np.cos(x[0]), 1.0 - 0.1 * dt]]), np.array([[0.0], [dt]])

    c = lambda k, x, u: 0.5 * (x @ x) + 0.05 * (u @ u) + 0.01 * k * x[0]                      # (time-dependent cost: exercises the k argument)
    cd = lambda k, x, u: (x + np.array([0.01 * k, 0.0]), np.eye(2), 0.1 * u, 0.1 * np.eye(1), np.zeros((1, 2)))
    h = lambda x: 2.0 * (x @ x)
    hd = lambda x: (4.0 * x, 4.0 * np.eye(2))
    W = lambda k: (1e-3 + 1e-4 * k) * np.eye(2)                                              # time-varying noise
    gen = rat.GenericRiskSensitiveProblem(f, c, h, W, N, n, m, f_returns_jacobian=True, c_derivatives=cd, h_derivatives=hd)
    cp = orc.ClosureProblem(lambda x, u: f(x, u), c, h, W, N, n, m, lambda x, u: f(x, u, True)[1:], cd, hd)
    return gen, cp, np.array([1.0, 0.0]), np.zeros((N, m))


def check_against_oracle(r, s, x, l, L, v, hist, vt=1e-9):
    assert r["status"] == 0 and r["iters"] == s.iter_current and r["ls_evals"] == len(hist)
    assert [e[0] for e in hist] == [e[0] for e in r["eps_history"]]                          # identical accepted / rejected step sizes
    assert abs(v - r["value"]) <= vt * abs(r["value"])
    assert rel(x, r["x"]) < 1e-9 and rel(l, r["l"]) < 1e-9 and rel(L, r["L"]) < 1e-9


@pytest.mark.parametrize("theta", [0.0, 0.5, 1.5])
def test_pendulum_closure_solve_against_the_oracle_closure_path(theta):
    gen, cp, x0, u0 = pendulum()
    s = rat.ILEQGSolver(gen, f_returns_jacobian=True)
    x, l, L, v, hist = rat.solve_(s, gen, x0, u0, theta=theta)
    r = orc.closure_solve(cp, x0, u0, theta)
    check_against_oracle(r, s, x, l, L, v, hist)
    assert theta == 0.0 or r["ls_evals"] > r["iters"]                      # the line search backtracks on this problem (theta > 0)
    # above the breakdown both refuse in initialize! (the reference's uncaught @assert, ileqg.jl:234 -> :440)
    assert orc.closure_solve(cp, x0, u0, 3.0)["status"] == 1
    with pytest.raises(AssertionError):
        rat.solve_(rat.ILEQGSolver(gen, f_returns_jacobian=True), gen, x0, u0, theta=3.0)


def test_f_returns_jacobian_problem_against_the_oracle_closure_path():
    """The family written as closures whose f returns its Jacobians (ileqg.jl:302-311), solved through the generic path and checked
    against the oracle's closure path AND the oracle's family path."""
    prob, x0, u = small_lq()
    f, c, cd, h, hd = lq_closures(prob)
    gen = rat.GenericRiskSensitiveProblem(f, c, h, prob.W, prob.N, prob.n, prob.m, f_returns_jacobian=True, c_derivatives=cd, h_derivatives=hd)
    cp = orc.ClosureProblem(lambda x, uu: f(x, uu), c, h, prob.W, prob.N, prob.n, prob.m, lambda x, uu: f(x, uu, True)[1:], cd, hd)
    for theta in (0.0, 1.5, 6.0):
        s = rat.ILEQGSolver(gen, f_returns_jacobian=True, adaptive_eps_init=True)
        x, l, L, v, hist = rat.solve_(s, gen, x0, u, theta=theta)
        check_against_oracle(orc.closure_solve(cp, x0, u, theta, adaptive_eps_init=True), s, x, l, L, v, hist)
        so = orc.ILEQGSolver(orc.Problem(prob), adaptive_eps_init=1)
        assert so.solve(x0, u, theta) == 0 and abs(v - so.s.value_current) <= 1e-9 * abs(v) and so.s.iter_current == s.iter_current


def test_closure_batch_solve_equals_per_sample_solves_and_the_oracle():
    """rat_dp_gain_sweep_batch / rat_dp_policy_eval_batch: a CE batch of a closure problem with every sweep of the batch in one launch."""
    gen, cp, x0, u0 = pendulum()
    theta = np.array([0.0, 0.2, 0.5, 1.0, 1.5, 1.7, 1e4])
    val, st, it, ls = rat.solve_closure_batch(gen, x0, u0, theta)
    assert st[-1] == 1 and np.isposinf(val[-1]) and np.all(st[:5] == 0)
    for i, th in enumerate(theta):
        r = orc.closure_solve(cp, x0, u0, th)
        assert r["status"] == st[i], (th, r["status"], st[i])
        if st[i] != 0:
            assert np.isposinf(val[i])
            continue
        assert r["iters"] == it[i] and r["ls_evals"] == ls[i]
        assert abs(val[i] - r["value"]) <= 1e-9 * abs(r["value"])
        s = rat.ILEQGSolver(gen, f_returns_jacobian=True)
        v1 = rat.solve_(s, gen, x0, u0, theta=th)[3]
        assert abs(val[i] - v1) <= 1e-12 * abs(v1) and s.iter_current == it[i]
    # a CE compute_cost over a closure problem goes through the same batch path and equals the device family's costs
    prob, lx0, lu = small_lq()
    f, c, cd, h, hd = lq_closures(prob)
    lgen = rat.GenericRiskSensitiveProblem(f, c, h, prob.W, prob.N, prob.n, prob.m, f_returns_jacobian=True, c_derivatives=cd, h_derivatives=hd)
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=6)
    th = np.array([0.1, 0.5, 1.0, 2.0, 5.0, 1e5])
    cg = rat.compute_cost(solver, lgen, lx0, lu, th, 0.3)
    cf = rat.compute_cost(rat.CrossEntropyBilevelOptimizationSolver(num_samples=6), prob, lx0, lu, th, 0.3)
    assert np.array_equal(np.isinf(cg), np.isinf(cf)) and np.isinf(cg[-1])
    assert np.all(np.abs(cg[:-1] - cf[:-1]) <= 1e-9 * np.abs(cf[:-1]))
