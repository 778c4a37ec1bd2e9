"""Problems beyond the 12 + 4 tile of the MFMA kernels (wide.hip: n <= 32, m <= 32, LQ family) against the oracle: the reference takes
its dimensions from the arrays (ileqg.jl:229), so a user of it can bring e.g. n = 20, m = 6.  Same bar as the tile-sized path: identical
status / iteration / line-search counts, values to 1e-9 relative, x / l / L of single solves to 1e-9 (1 + |.|), CE on an injected stream."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from ratilqr.jl_amd import nelder_mead as nm
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
VT = 1e-9


def wide_problem(n, m, N, seed, kappa=0.0, tv=False, w=2e-3):
    """random well-posed LQ problem: stable-ish A, SPD Hessians, cross terms, linear terms, non-diagonal (optionally time-varying) W"""
    g = np.random.default_rng(seed)
    Qo, _ = np.linalg.qr(g.standard_normal((n, n)))
    A = 0.92 * Qo
    B = g.standard_normal((n, m)) / np.sqrt(n)

    def spd(k, lo, T=None):
        def one():
            X = g.standard_normal((k, k))
            return lo * np.eye(k) + X @ X.T / k
        return np.stack([one() for _ in range(T)]) if T else one()

    T = N if tv else None
    Q, R = spd(n, 0.5, T), spd(m, 0.2, T)
    P = 0.05 * g.standard_normal((N, m, n) if tv else (m, n))
    qv = 0.1 * g.standard_normal((N, n) if tv else n)
    rv = 0.1 * g.standard_normal((N, m) if tv else m)
    q0 = g.standard_normal(N) if tv else 0.3
    W = w * spd(n, 0.5, T)
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Q, R=R, N=N, W=W, P=P, qv=qv, rv=rv, q0=q0, Qf=spd(n, 0.5), qvf=0.1 * g.standard_normal(n),
                                      q0f=0.7, kappa=kappa)
    return prob, g.standard_normal(n), 0.05 * g.standard_normal((N, m))


def theta_grid(P, x0, u, k=10):
    """theta = 0, a ladder below the breakdown of initialize! (bisected with the oracle) and two infeasible values"""
    lo, hi = 0.0, 1.0
    while orc.compute_value_batch(P, x0, u, np.array([hi]))[1][0] == 0 and hi < 1e6:
        lo, hi = hi, 2 * hi
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        if orc.compute_value_batch(P, x0, u, np.array([mid]))[1][0] in (0, 3):
            lo = mid
        else:
            hi = mid
    return np.concatenate([[0.0], lo * np.linspace(0.05, 0.97, k), [1.05 * hi, 3.0 * hi]])


def check_batch(ctx, P, x0, u, theta, **opts):
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta, nthreads=8, **opts)
    vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
    assert np.array_equal(sg, so), (sg, so)
    assert np.array_equal(ig, io) and np.array_equal(lg, lo), (ig, io, lg, lo)
    fin = np.isfinite(vo)
    assert np.array_equal(fin, np.isfinite(vg)) and np.all(np.isposinf(vg[~fin]))
    assert np.all(np.abs(vg[fin] - vo[fin]) <= VT * np.abs(vo[fin])), np.abs(vg[fin] / vo[fin] - 1).max()
    return vg, sg


@pytest.mark.parametrize("n,m,N,seed,kappa,tv", [
    (13, 4, 20, 1, 0.0, False),        # one state beyond the tile
    (8, 6, 25, 2, 0.0, False),         # only the controls exceed it
    (20, 6, 30, 3, 0.0, True),         # time-varying cost tables and W(k)
    (16, 5, 20, 4, 0.02, False),       # cubic drift: several iterations, backtracking
    (32, 8, 12, 5, 0.0, False),        # the largest state dimension
    (24, 24, 10, 6, 0.0, True),        # m = n
    (32, 32, 6, 7, 0.0, False),        # the largest of both (101 KB of LDS per sample)
])
def test_batched_solves_match_the_oracle(n, m, N, seed, kappa, tv):
    prob, x0, u = wide_problem(n, m, N, seed, kappa, tv)
    P = orc.Problem(prob)
    theta = theta_grid(P, x0, u)
    ctx = rat.Context(prob, max_batch=theta.size)
    vg, sg = check_batch(ctx, P, x0, u, theta)
    assert sg[0] == 0 and (sg[-2:] != 0).all() and (sg[1:-2] == 0).sum() >= 5
    ctx.profile(True)
    ctx.solve_batch(x0, u, theta)
    kinds = [k for k, v in ctx.profile_get().items() if v["launches"]]
    assert kinds == ["solve_wide"], kinds


def test_backtracking_line_searches():
    """cubic drift: rejected step sizes (more evaluations than iterations)"""
    for n, m, N, seed, kappa in ((14, 6, 30, 0, 0.05), (16, 5, 20, 1, 0.05), (14, 6, 30, 6, 0.04)):
        prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=seed, kappa=kappa)
        theta = np.array([0.0, 1.0, 3.0, 6.0])
        ctx = rat.Context(prob, max_batch=4)
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=4)
        assert (lo > io).any()
        check_batch(ctx, orc.Problem(prob), x0, u, theta)


def test_single_solves_return_trajectory_policy_and_history():
    prob, x0, u = wide_problem(18, 5, 25, 11, kappa=0.015)
    P = orc.Problem(prob)
    ctx = rat.Context(prob)
    th = theta_grid(P, x0, u, 4)
    for theta in (0.0, th[2], th[4]):
        so = orc.ILEQGSolver(P)
        assert so.solve(x0, u, theta) == 0
        r = ctx.solve(x0, u, theta)
        assert r["status"] == 0 and r["iters"] == so.s.iter_current and r["hist_n"] == so.s.n_hist
        assert abs(r["value"] - so.s.value_current) <= VT * abs(so.s.value_current)
        for k, ref in (("x", so.x_array), ("l", so.l_array), ("L", so.L_array)):
            assert np.abs(r[k] - ref).max() <= VT * (1 + np.abs(ref).max()), k
        assert np.array_equal(r["eps_history"][:, 0], so.eps_history[:, 0])
        assert np.abs(r["eps_history"][:, 1] - so.eps_history[:, 1]).max() <= 1e-7 * (1 + np.abs(so.eps_history[:, 1]).max())


def test_options_and_iter_max():
    prob, x0, u = wide_problem(14, 5, 15, 21, kappa=0.03)
    P = orc.Problem(prob)
    th = theta_grid(P, x0, u, 6)
    for kw in (dict(iter_max=1), dict(iter_max=2, d=1e-6), dict(adaptive_eps_init=True, eps_init=0.5, lam=0.7),
               dict(mu_min=1e-3, Delta_0=3.0, d=1e-4)):
        ctx = rat.Context(prob, rat.ileqg.make_opts(**kw), max_batch=th.size)
        okw = {{"Delta_0": "delta_0"}.get(k, k): (int(v) if k == "adaptive_eps_init" else v) for k, v in kw.items()}
        check_batch(ctx, P, x0, u, th, **okw)


def test_handle_switches_between_tile_sized_and_wide_problems():
    small, sx0, su = rat.synthetic_lq_problem(n=6, m=2, N=12, seed=3)
    big, bx0, bu = wide_problem(15, 3, 12, 31)
    th = np.array([0.0, 0.4, 0.9])
    ctx = rat.Context(small, max_batch=3)
    a = check_batch(ctx, orc.Problem(small), sx0, su, th)[0]
    ctx.set_problem(big)
    thb = theta_grid(orc.Problem(big), bx0, bu, 3)[:3]
    check_batch(ctx, orc.Problem(big), bx0, bu, thb)
    ctx.set_problem(small)
    assert np.array_equal(check_batch(ctx, orc.Problem(small), sx0, su, th)[0], a)


def test_operators_fail_loudly_beyond_the_tile():
    prob, x0, u = wide_problem(13, 2, 8, 41)
    ctx = rat.Context(prob)
    with pytest.raises(rat.native.RatError, match="n <= 12"):
        ctx.rollout_open(x0, u)
    with pytest.raises(rat.native.RatError, match="n <= 32"):
        rat.Context(rat.LQRiskSensitiveProblem(np.eye(40), np.ones((40, 2)), Q=np.eye(40), R=np.eye(2), N=5, W=np.eye(40)))


def test_ce_solve_matches_the_oracle_on_an_injected_stream():
    prob, x0, u = wide_problem(16, 6, 15, 51)
    P = orc.Problem(prob)
    hi = theta_grid(P, x0, u, 2)[-2] / 1.05
    z = np.random.default_rng(777).standard_normal(20000)
    kw = dict(num_samples=24, num_elite=5, iter_max=3, mu_init=0.4 * hi, sigma_init=0.4 * hi)
    solver = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    got = ce.solve_(solver, prob, x0, u, z, kl_bound=0.2)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, **kw)
    rc, th, x, l, L, val, tmin, tmax = oc.solve(P, x0, u, 0.2)
    assert rc == 0
    assert abs(got[0] - th) <= 1e-9 * abs(th) and abs(got[4] - val) <= 1e-9 * abs(val)
    assert got[5] == tmin and got[6] == tmax
    assert solver.c.mu_init == oc.c.mu_init and solver.c.sigma_init == oc.c.sigma_init
    assert solver.c.n_solves == oc.c.n_solves and solver.c.n_redraws == oc.c.n_redraws
    assert np.abs(got[1] - x).max() < 1e-9 * (1 + np.abs(x).max()) and np.abs(got[3] - L).max() < 1e-9 * (1 + np.abs(L).max())


def test_value_against_the_first_principles_gaussian_integral():
    """the wide sweep's own algebra (D S = S + theta Z'Z, logdet by Cholesky) against tests/leqg_exact.py: no Riccati recursion there"""
    import leqg_exact as ex
    prob, x0, u = wide_problem(14, 5, 10, 61, tv=True)
    P = orc.Problem(prob)
    th = theta_grid(P, x0, u, 4)
    ctx = rat.Context(prob)
    for theta in th[1:5]:
        r = ctx.solve(x0, u, theta)
        assert r["status"] == 0
        exact, feasible = ex.exact_value(prob, x0, r["l"], None, r["L"], r["x"], theta)
        assert feasible and abs(r["value"] - exact) <= 1e-9 * abs(exact), (theta, r["value"], exact)


def test_nelder_mead_matches_the_sequential_oracle():
    """RAT iLQR++ on a 16-state problem; theta_high_init beyond the breakdown is halved until feasible (nm.jl:283-293)"""
    prob, x0, u = wide_problem(16, 6, 15, 51)
    P = orc.Problem(prob)
    hi = theta_grid(P, x0, u, 2)[-2] / 1.05
    kw = dict(theta_high_init=6.0 * hi, theta_low_init=1e-8, iter_max=12, eps=1e-4)
    so, sg = orc.NelderMeadBilevelOptimizationSolver(**kw), rat.NelderMeadBilevelOptimizationSolver(**kw)
    rc, th_o, x_o, l_o, L_o, v_o = so.solve(P, x0, u, 0.2)
    th_g, x_g, l_g, L_g, v_g = nm.solve_(sg, prob, x0, u, kl_bound=0.2)
    assert rc == 0 and so.c.iter_current == sg.c.iter_current and so.c.n_solves == sg.c.n_solves
    assert abs(th_g - th_o) <= 1e-9 * abs(th_o) and abs(v_g - v_o) <= 1e-9 * abs(v_o)
    assert sg.c.theta_high_init == so.c.theta_high_init < 6.0 * hi
    assert np.abs(x_g - x_o).max() < 1e-8 * (1 + np.abs(x_o).max()) and np.abs(L_g - L_o).max() < 1e-8 * (1 + np.abs(L_o).max())


def test_multi_device_object_takes_wide_problems(monkeypatch):
    monkeypatch.setenv("RATILQR_MULTI_FORCE_RCCL", "1")
    prob, x0, u = wide_problem(20, 6, 12, 71)
    theta = theta_grid(orc.Problem(prob), x0, u, 20)
    mc = rat.MultiContext(prob, max_batch=theta.size, devices=(0,))
    cost = mc.compute_cost(x0, u, theta, 0.1)
    v, st, _, _ = rat.Context(prob, max_batch=theta.size).solve_batch(x0, u, theta)
    with np.errstate(divide="ignore"):
        want = np.where(np.isfinite(v), v + 0.1 / theta, np.inf)
    assert np.array_equal(cost, want) and mc.allgathers == 1
