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
    (1, 5, 10, 8, 0.01, False),        # more controls than states: m > n (the scratch of the lockstep factorisation once assumed n >= m)
    (3, 32, 6, 9, 0.0, True),
    (1, 32, 1, 10, 0.0, False),        # a single step, a single state
    (32, 1, 8, 11, 0.01, True),
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


@pytest.mark.parametrize("n,m,N,seed,kappa,tv", [
    (16, 4, 30, 21, 0.0, False),       # the full tile of the register sweep (wide16.h)
    (16, 4, 20, 22, 0.03, True),       # cubic drift, time-varying cost tables and W(k)
    (13, 1, 25, 23, 0.0, False),       # odd n: the last pivot block pairs a state with a padded unit row; one control
    (15, 3, 20, 24, 0.02, False),
    (14, 2, 12, 25, 0.0, True),
    (16, 4, 1, 26, 0.0, False),        # a single step
])
def test_register_sweep_sizes_match_the_oracle_and_the_general_sweep(n, m, N, seed, kappa, tv, monkeypatch):
    """n <= 16, m <= 4 beyond the 12 + 4 tile: the solve kernel's sweeps run in registers on the matrix pipe (wide16.h: two-tile form of the
    tile kernels' recursion, elimination with 2 x 2 block pivots instead of Cholesky factors).  Same bar against the oracle as every wide size,
    and the general LDS sweep (switch wide16 = 0) must agree on every count and to 1e-10 on the values."""
    prob, x0, u = wide_problem(n, m, N, seed, kappa, tv)
    P = orc.Problem(prob)
    theta = theta_grid(P, x0, u)
    ctx = rat.Context(prob, max_batch=theta.size)
    assert ctx.debug_get("wide16") == 1
    vg, sg = check_batch(ctx, P, x0, u, theta)
    assert sg[0] == 0 and (sg[-2:] != 0).all() and (sg[1:-2] == 0).sum() >= 5
    _, _, ig, lg = ctx.solve_batch(x0, u, theta)
    monkeypatch.setenv("RATILQR_WIDE16", "0")
    ref = rat.Context(prob, max_batch=theta.size)
    monkeypatch.delenv("RATILQR_WIDE16")
    assert ref.debug_get("wide16") == 0
    vr, sr, ir, lr = ref.solve_batch(x0, u, theta)
    assert np.array_equal(sg, sr) and np.array_equal(ig, ir) and np.array_equal(lg, lr)
    fin = np.isfinite(vr)
    assert np.all(np.abs(vg[fin] - vr[fin]) <= 1e-10 * np.abs(vr[fin]))
    # single solve: trajectory, controls and gains
    th = float(theta[3])
    r1, r0 = ctx.solve(x0, u, th), ref.solve(x0, u, th)
    assert rel(r1["x"], r0["x"]) < 1e-10 and rel(r1["l"], r0["l"]) < 1e-10 and rel(r1["L"], r0["L"]) < 1e-9
    so = orc.ILEQGSolver(P)
    assert so.solve(x0, u, th) == 0 and r1["status"] == 0 and r1["iters"] == so.s.iter_current
    for k, o in (("x", so.x_array), ("l", so.l_array), ("L", so.L_array)):
        assert np.abs(r1[k] - o).max() <= VT * (1 + np.abs(o).max()), k


@pytest.mark.parametrize("n,m,N,seed,kappa,tv", [
    (20, 6, 30, 31, 0.0, False),       # two state tiles, one control tile
    (24, 8, 20, 32, 0.02, True),       # cubic drift, time-varying cost tables and W(k)
    (32, 32, 8, 33, 0.0, False),       # two by two: every tile full
    (17, 17, 12, 34, 0.01, True),      # one row / column into the second tiles: the odd pivot block pairs a state with a padded unit row
    (8, 6, 25, 35, 0.0, False),        # only the controls exceed the 12 + 4 tile: one tile each
    (5, 20, 10, 36, 0.0, False),       # few states, two control tiles
    (32, 1, 10, 37, 0.02, False),
])
def test_block_form_sizes_match_the_oracle_and_the_general_sweep(n, m, N, seed, kappa, tv, monkeypatch):
    """Every general size beyond wide16.h's runs its sweeps and rollouts in registers on blocks of 16 x 16 tiles (wide32.h: elimination with
    2 x 2 block pivots over the blocks for M and for H, tables as register images).  Same bar against the oracle as every wide size, and the
    general LDS sweep (switch wide32 = 0) must agree on every count and to 1e-10 on the values; single solve: trajectory, controls, gains."""
    prob, x0, u = wide_problem(n, m, N, seed, kappa, tv)
    P = orc.Problem(prob)
    theta = theta_grid(P, x0, u)
    ctx = rat.Context(prob, max_batch=theta.size)
    assert ctx.debug_get("wide32") == 1 and ctx.debug_get("wide16") == 0
    vg, sg = check_batch(ctx, P, x0, u, theta)
    assert sg[0] == 0 and (sg[-2:] != 0).all() and (sg[1:-2] == 0).sum() >= 5
    _, _, ig, lg = ctx.solve_batch(x0, u, theta)
    monkeypatch.setenv("RATILQR_WIDE32", "0")
    ref = rat.Context(prob, max_batch=theta.size)
    monkeypatch.delenv("RATILQR_WIDE32")
    assert ref.debug_get("wide32") == 0
    vr, sr, ir, lr = ref.solve_batch(x0, u, theta)
    assert np.array_equal(sg, sr) and np.array_equal(ig, ir) and np.array_equal(lg, lr)
    fin = np.isfinite(vr)
    assert np.all(np.abs(vg[fin] - vr[fin]) <= 1e-10 * np.abs(vr[fin]))
    th = float(theta[3])
    r1, r0 = ctx.solve(x0, u, th), ref.solve(x0, u, th)
    assert rel(r1["x"], r0["x"]) < 1e-10 and rel(r1["l"], r0["l"]) < 1e-10 and rel(r1["L"], r0["L"]) < 1e-9
    so = orc.ILEQGSolver(P)
    assert so.solve(x0, u, th) == 0 and r1["status"] == 0 and r1["iters"] == so.s.iter_current
    for k, o in (("x", so.x_array), ("l", so.l_array), ("L", so.L_array)):
        assert np.abs(r1[k] - o).max() <= VT * (1 + np.abs(o).max()), k


def test_block_form_regularisation_restarts():
    """wide32.h: an indefinite c_uu -- the elimination of H meets a non-positive leading minor, mu and Delta are raised and the sweep restarts
    (ileqg.jl:372-378) -- at a size with two control tiles"""
    prob, x0, u = wide_problem(18, 20, 8, 92)
    prob.R = prob.R - 0.9 * np.eye(20) * np.linalg.eigvalsh(prob.R).max()
    P = orc.Problem(prob)
    theta = np.array([0.0, 0.05, 0.2])
    ctx = rat.Context(prob, max_batch=3)
    assert ctx.debug_get("wide32") == 1
    check_batch(ctx, P, x0, u, theta)


def test_register_sweep_regularisation_restarts_and_backtracking():
    """wide16.h: an indefinite c_uu (H loses positive definiteness: mu, Delta raised, the sweep restarts, ileqg.jl:372-378) and cubic drift
    with rejected step sizes"""
    prob, x0, u = wide_problem(14, 4, 10, 91)
    prob.R = prob.R - 0.9 * np.eye(4) * np.linalg.eigvalsh(prob.R).max()
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=3)
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, np.array([0.0, 0.1, 0.3]), nthreads=3)
    check_batch(ctx, P, x0, u, np.array([0.0, 0.1, 0.3]))
    for n, m, N, seed, kappa in ((16, 4, 20, 3, 0.05), (14, 3, 30, 2, 0.05), (13, 2, 30, 0, 0.05)):
        prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=seed, kappa=kappa)
        theta = np.array([0.0, 1.0, 3.0, 6.0])
        ctx = rat.Context(prob, max_batch=4)
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=4)
        assert (lo > io).any()
        check_batch(ctx, orc.Problem(prob), x0, u, theta)


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


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_sizes_beyond_the_general_kernel_fail_loudly():
    with pytest.raises(rat.native.RatError, match="n <= 32"):
        rat.Context(rat.LQRiskSensitiveProblem(np.eye(40), np.ones((40, 2)), Q=np.eye(40), R=np.eye(2), N=5, W=np.eye(40)))
    with pytest.raises(rat.native.RatError, match="LQ family"):
        rat.Context(rat.PowerLawRiskSensitiveProblem(13, 5, 0.01 * np.eye(13)))


@pytest.mark.parametrize("n,m,N,seed,kappa,tv", [(14, 5, 12, 81, 0.03, True), (32, 9, 6, 82, 0.0, False), (7, 7, 9, 83, 0.01, True),
                                                 (2, 7, 5, 84, 0.0, False), (3, 32, 4, 85, 0.02, True), (32, 32, 3, 86, 0.0, False)])
def test_operator_forms_match_the_oracle(n, m, N, seed, kappa, tv):
    """simulate_dynamics (three forms), integrate_cost, approximate_model, solve_approximate_dp(!) as individual calls at general size"""
    prob, x0, u = wide_problem(n, m, N, seed, kappa, tv)
    P = orc.Problem(prob)
    ctx = rat.Context(prob)
    _, xo = orc.simulate_open(P, x0, u)
    xg = ctx.rollout_open(x0, u)
    assert rel(xg, xo) < 1e-12
    assert abs(ctx.integrate_cost(xo, u) - orc.integrate_cost(P, xo, u)[1]) <= 1e-12 * abs(orc.integrate_cost(P, xo, u)[1])
    ap = ctx.approximate_model(u, xo)
    _, ap_o = orc.approximate_model(P, u, xo)
    a = ap_o.arrays()
    for kk, name in (("q", "q_array"), ("qv", "q_vec_array"), ("Q", "Q_array"), ("r", "r_array"), ("R", "R_array"),
                     ("P", "P_array"), ("A", "A_array"), ("B", "B_array"), ("W", "W_array")):
        assert rel(getattr(ap, name), a[kk]) < 1e-12, kk
    th_hi = theta_grid(P, x0, u, 2)[2]
    for theta in (0.0, 0.6 * th_hi):
        st, L, dl, dp, mu1, de1 = ctx.dp_gain_sweep(ap, theta, 0.0, 2.0)
        _, Lo, dlo, dpo, mu_o, de_o = orc.dp_gain(P, ap_o, theta)
        assert st == 0 and mu1 == mu_o and de1 == de_o
        for got, ref in ((L, Lo), (dl, dlo), (dp.s_array, dpo["s"]), (dp.S_array, dpo["S"]), (dp.s_vec_array, dpo["sv"]),
                         (dp.g_array, dpo["g"]), (dp.G_array, dpo["G"]), (dp.H_array, dpo["H"])):
            assert rel(got, ref) < 1e-9
        # the policy just computed: closed-loop rollout, then its evaluation with and without dl
        _, xn_o, un_o = orc.simulate_feedback(P, xo, u + dlo, Lo)
        xn, un = ctx.rollout_feedback(xo, u + dlo, Lo)
        assert rel(xn, xn_o) < 1e-12 and rel(un, un_o) < 1e-12
        for dl_in in (None, dlo):
            st2, dp2 = ctx.dp_policy_eval(ap, Lo, dl_in, theta, 0.0)
            _, dpo2 = orc.dp_eval(P, ap_o, Lo, dl_in, theta, 0.0)
            assert st2 == 0 and rel(dp2.s_array, dpo2["s"]) < 1e-9 and rel(dp2.S_array, dpo2["S"]) < 1e-9 and rel(dp2.s_vec_array, dpo2["sv"]) < 1e-9
        assert abs(dp2.s_array[0] - dp.s_array[0]) <= 1e-9 * abs(dp.s_array[0])       # K8: evaluating (L, dl) returns the gain sweep's value
    big = 1e3 * th_hi
    assert ctx.dp_gain_sweep(ap, big, 0.0, 2.0)[0] == orc.dp_gain(P, ap_o, big)[0] == 2   # @assert isposdef(M)
    # Monte-Carlo rollouts under process noise, injected draws: open loop and under the affine policy
    z = np.random.default_rng(5).standard_normal((6, N, n))
    for Lpol in (None, Lo):
        xnom = x0 if Lpol is None else xo
        _, xz_o, uz_o, cz_o = orc.simulate_noisy(P, xnom, u, Lpol, z)
        xz, uz, cz, dom = ctx.rollout_noisy(xnom, u, Lpol, z=z)
        assert not dom and rel(xz, xz_o) < 1e-12 and rel(uz, uz_o) < 1e-12 and rel(cz, cz_o) < 1e-12
    _, _, c1, _ = ctx.rollout_noisy(x0, u, None, K=2000, seed=9, want_x=False, want_u=False)   # device generator: reproducible, sane moments
    _, _, c2, _ = ctx.rollout_noisy(x0, u, None, K=2000, seed=9, want_x=False, want_u=False)
    assert np.array_equal(c1, c2) and np.all(np.isfinite(c1))
    if kappa == 0.0:                                                                    # (linear dynamics: the quadratic model is exact)
        assert abs(c1.mean() - ctx.dp_policy_eval(ap, np.zeros((N, m, n)), None, 0.0, 0.0)[1].s_array[0]) < 6 * c1.std() / np.sqrt(2000)


def test_regularisation_restarts_inside_the_gain_sweep():
    """an indefinite c_uu makes H lose positive definiteness: mu, Delta are raised and the sweep restarts (ileqg.jl:372-378)"""
    prob, x0, u = wide_problem(14, 5, 10, 91)
    prob.R = prob.R - 0.9 * np.eye(5) * np.linalg.eigvalsh(prob.R).max()
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=3)
    _, xo = orc.simulate_open(P, x0, u)
    ap, (_, ap_o) = ctx.approximate_model(u, xo), orc.approximate_model(P, u, xo)
    st, L, dl, dp, mu1, de1 = ctx.dp_gain_sweep(ap, 0.1, 0.0, 2.0)
    rc, Lo, dlo, dpo, mu_o, de_o = orc.dp_gain(P, ap_o, 0.1)
    assert st == rc == 0 and mu1 == mu_o > 0 and de1 == de_o and rel(L, Lo) < 1e-9
    check_batch(ctx, P, x0, u, np.array([0.0, 0.1, 0.3]))


def test_stepwise_composition_equals_the_single_launch_solve():
    """initialize! / step! composed through the operator ABI == the solve kernel (same statements, different entry points)"""
    prob, x0, u = rat.synthetic_lq_problem(n=14, m=6, N=30, seed=0, kappa=0.05)
    for theta in (0.0, 3.0):
        s1, s2 = rat.ILEQGSolver(prob), rat.ILEQGSolver(prob)
        x1, l1, L1, v1, h1 = rat.solve_(s1, prob, x0, u, theta=theta)
        x2, l2, L2, v2, h2 = rat.solve_stepwise_(s2, prob, x0, u, theta)
        assert s1.iter_current == s2.iter_current and [a[0] for a in h1] == [a[0] for a in h2]
        assert abs(v1 - v2) <= 1e-11 * abs(v2) and rel(x1, x2) < 1e-11 and rel(L1, L2) < 1e-10


def test_closure_problems_at_general_size():
    """the LQ family written as host closures whose f returns its Jacobians (ileqg.jl:302-311), n = 14: host rollouts + linearisation,
    device sweeps through rat_dp_* and their batch forms, against the oracle's closure path and the device family"""
    rng = np.random.default_rng(3)
    n, m, N = 14, 5, 10
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    prob = rat.LQRiskSensitiveProblem(0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), Q=np.eye(n), R=0.3 * np.eye(m),
                                      P=0.05 * rng.standard_normal((m, n)), qv=0.1 * rng.standard_normal(n), rv=0.1 * rng.standard_normal(m),
                                      q0=0.2, N=N, W=1e-3 * np.eye(n), Qf=np.eye(n), kappa=0.02)
    x0, u = 0.5 * rng.standard_normal(n), np.zeros((N, m))
    A, B, kap, Q, R, Pm, qv, rv, q0, Qf = prob.A, prob.B, prob.kappa, prob.Q, prob.R, prob.P, prob.qv, prob.rv, float(prob.q0), prob.Qf

    def f(x, uu, f_returns_jacobian=False):
        xn = A @ x + B @ uu + kap * x ** 3
        return (xn, A + np.diag(3 * kap * x ** 2), B) if f_returns_jacobian else xn
    c = lambda k, x, uu: 0.5 * x @ Q @ x + 0.5 * uu @ R @ uu + uu @ Pm @ x + qv @ x + rv @ uu + q0       # noqa: E731
    cd = lambda k, x, uu: (Q @ x + Pm.T @ uu + qv, Q, R @ uu + Pm @ x + rv, R, Pm)                      # noqa: E731
    h = lambda x: 0.5 * x @ Qf @ x                                                                      # noqa: E731
    hd = lambda x: (Qf @ x, Qf)                                                                         # noqa: E731
    gen = rat.GenericRiskSensitiveProblem(f, c, h, prob.W, N, n, m, f_returns_jacobian=True, c_derivatives=cd, h_derivatives=hd)
    cp = orc.ClosureProblem(lambda x, uu: f(x, uu), c, h, prob.W, N, n, m, lambda x, uu: f(x, uu, True)[1:], cd, hd)
    theta = np.array([0.0, 1.0, 4.0, 1e5])
    val, st, it, ls = rat.solve_closure_batch(gen, x0, u, theta)
    vf, sf, itf, lsf = rat.Context(prob, max_batch=4).solve_batch(x0, u, theta)
    assert np.array_equal(st, sf) and np.array_equal(it, itf) and np.array_equal(ls, lsf) and st[-1] == 1
    assert np.all(np.abs(val[:3] - vf[:3]) <= 1e-9 * np.abs(vf[:3]))
    for i in range(3):
        r = orc.closure_solve(cp, x0, u, theta[i])
        assert r["status"] == 0 and r["iters"] == it[i] and abs(val[i] - r["value"]) <= 1e-9 * abs(r["value"])


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


def test_full_size_ce_batch_at_general_size():
    """1024 theta-samples at n = 20, m = 6, N = 50: the oracle on a sample of the batch, size-independent properties on all of it"""
    prob, x0, u = rat.synthetic_lq_problem(n=20, m=6, N=50)
    P = orc.Problem(prob)
    hi = theta_grid(P, x0, u, 2)[-2] / 1.05
    rng = np.random.default_rng(5)
    theta = np.sort(np.concatenate([rng.uniform(0.0, 0.98 * hi, 1000), rng.uniform(1.02 * hi, 3.0 * hi, 24)]))
    ctx = rat.Context(prob, max_batch=1024)
    v, st, it, ls = ctx.solve_batch(x0, u, theta)
    feas = theta < hi
    assert np.all(st[feas] == 0) and np.all(st[~feas] == 1) and np.all(np.isposinf(v[~feas]))
    assert np.all(np.diff(v[feas]) > 0)                                   # the value grows with the risk-sensitivity parameter
    pick = np.concatenate([np.arange(0, 1000, 37), np.arange(1000, 1024, 5)])
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta[pick], nthreads=8)
    assert np.array_equal(so, st[pick]) and np.array_equal(io, it[pick]) and np.array_equal(lo, ls[pick])
    f = np.isfinite(vo)
    assert np.all(np.abs(v[pick][f] - vo[f]) <= VT * np.abs(vo[f]))
    v2, st2, _, _ = ctx.solve_batch(x0, u, theta[::-1].copy())           # batch composition / order must not matter
    assert np.array_equal(v2[::-1], v) and np.array_equal(st2[::-1], st)
