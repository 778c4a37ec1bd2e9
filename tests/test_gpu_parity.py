"""GPU parity of the batched iLEQG hot path against the CPU oracle and the committed golden vectors.

Tolerances: values 1e-9 relative, trajectories/gains 1e-9*(1+|.|inf) absolute, identical
status / iteration count / line-search count, identical results for every speculation width E."""
import json
import os

import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
VT = 1e-9


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def check_batch(ctx, P, x0, u, theta, **opts):
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta, nthreads=8, **opts)
    vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
    assert np.array_equal(sg, so), (sg, so)
    assert np.array_equal(ig, io) and np.array_equal(lg, lo), (ig, io, lg, lo)
    fin = np.isfinite(vo)
    assert np.array_equal(fin, np.isfinite(vg))
    assert np.all(np.isposinf(vg[~fin]))
    if fin.any():
        assert np.all(np.abs(vg[fin] - vo[fin]) <= VT * np.abs(vo[fin]))
    return vg, sg, ig, lg


def test_config2_golden_256_thetas():
    """BASELINE config 2: 256 theta = linspace(0.01, 0.8 theta_max), N=50, n=12, m=4, one compute_cost batch."""
    g = GOLD["config2"]
    prob, x0, u = rat.synthetic_lq_problem()
    ctx = rat.Context(prob, max_batch=256, spec_eps=1)
    v, st, it, ls = ctx.solve_batch(x0, u, np.array(g["theta"]))
    assert np.array_equal(st, g["status"]) and np.array_equal(it, g["iters"]) and np.array_equal(ls, g["ls"])
    assert np.all(np.abs(v - np.array(g["value"])) <= VT * np.abs(g["value"]))
    assert np.all(np.diff(v) > 0)                  # value is increasing in theta on this problem
    # edges: theta = 0 (iLQG branch), just below / above the breakdown theta of initialize!
    e = GOLD["config2_edge"]
    v, st, _, _ = ctx.solve_batch(x0, u, np.array(e["theta"]))
    assert np.array_equal(st, e["status"])
    for a, b in zip(v, e["value"]):
        assert (b is None and np.isposinf(a)) or abs(a - b) <= VT * abs(b)


def test_first_gain_sweep_policy_against_oracle():
    """config 2: L, dl of the first gain sweep and the converged policy of single solves."""
    prob, x0, u = rat.synthetic_lq_problem()
    P = orc.Problem(prob)
    ctx = rat.Context(prob)
    for theta in (0.0, 0.3, 6.0, 10.0):
        so = orc.ILEQGSolver(P)
        assert so.solve(x0, u, theta) == 0
        r = ctx.solve(x0, u, theta)
        assert r["status"] == 0 and r["iters"] == so.s.iter_current and r["hist_n"] == so.s.n_hist
        assert abs(r["value"] - so.s.value_current) <= VT * abs(so.s.value_current)
        for k, ref in (("x", so.x_array), ("l", so.l_array), ("L", so.L_array)):
            assert np.abs(r[k] - ref).max() <= VT * (1 + np.abs(ref).max())
        assert np.array_equal(r["eps_history"][:, 0], so.eps_history[:, 0])


@pytest.mark.parametrize("E", [1, 2, 4, 8, 11])
def test_speculation_width_does_not_change_results(E):
    """App. B.17: evaluating E step sizes at once and replaying the sequential rule is result-identical."""
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    P = orc.Problem(prob)
    theta = np.array([0.0, 1.0, 2.0, 4.0, 5.0, 5.9, 6.3, 6.6, 9.0])
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=E)
    vg, sg, ig, lg = check_batch(ctx, P, x0, u, theta)
    g = GOLD["cubic_backtracking"]
    sel = [0, 2, 4]
    assert np.all(np.abs(vg[sel] - np.array(g["value"])[[0, 1, 2]]) <= VT * np.abs(np.array(g["value"])[[0, 1, 2]]))
    assert lg[4] == 6 and ig[4] == 4           # theta = 5: eps = 1, 1, 1, 1, 1/2, 1/4


def test_batch_composition_independence():
    """App. B.16: a sample's result must not depend on what else is in the batch (fresh solver per sample)."""
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    theta = np.array([0.0, 3.0, 5.0, 6.2, 30.0, 1.0, 6.45])
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=4)
    v_all, s_all, i_all, l_all = ctx.solve_batch(x0, u, theta)
    perm = np.random.default_rng(0).permutation(theta.size)
    v_p, s_p, i_p, l_p = ctx.solve_batch(x0, u, theta[perm])
    assert np.array_equal(v_all[perm], v_p) and np.array_equal(s_all[perm], s_p) and np.array_equal(l_all[perm], l_p)
    for k in range(theta.size):
        v1, s1, i1, l1 = ctx.solve_batch(x0, u, theta[k:k + 1])
        assert (v1[0] == v_all[k] or (np.isinf(v1[0]) and np.isinf(v_all[k]))) and s1[0] == s_all[k] and i1[0] == i_all[k]


def test_config1_small_problems_golden():
    """BASELINE config 1: N=20, n=4, m=2 LQ plumbing instance (padded into the 12/4 kernels)."""
    g = GOLD["config1_n4"]
    prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
    ctx = rat.Context(prob, max_batch=4, spec_eps=2)
    v, st, it, ls = ctx.solve_batch(x0, u, np.array(g["theta"]))
    assert np.array_equal(st, g["status"]) and np.array_equal(it, g["iters"])
    for a, b in zip(v, g["value"]):
        assert (b is None and np.isposinf(a)) or abs(a - b) <= VT * abs(b)
    r = ctx.solve(x0, u, 0.5)
    assert rel(r["L"][0], g["L0"]) < VT and rel(r["x"][-1], g["x_end"]) < 1e-8


def test_nonlinear_powerlaw_golden():
    g = GOLD["nonlinear_test"]
    prob = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2))
    ctx = rat.Context(prob, max_batch=5, spec_eps=3)
    theta = np.array([0.0, 0.1, 0.3, 0.43, 0.5])
    v, st, it, ls = ctx.solve_batch(np.zeros(2), 0.1 * np.ones((10, 2)), theta)
    for k, th in enumerate(theta):
        e = g[repr(float(th))]
        assert st[k] == e["rc"] and it[k] == e["iters"] and ls[k] == e["ls"] and abs(v[k] - e["value"]) <= VT * e["value"]


def test_domain_error_maps_to_status_and_inf():
    """x < 0 with a fractional exponent throws DomainError in the reference -> status 4, value Inf."""
    prob = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2))
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=2)
    x0, u = np.array([-0.2, 0.1]), 0.1 * np.ones((10, 2))
    vo, so, _, _ = orc.compute_value_batch(P, x0, u, [0.0, 0.3])
    vg, sg, _, _ = ctx.solve_batch(x0, u, np.array([0.0, 0.3]))
    assert np.array_equal(so, sg) and np.all(sg == 4) and np.all(np.isposinf(vg))


def stress_problem(seed, kappa=0.03, qs=-0.2, rw=0.1):
    rng = np.random.default_rng(seed)
    n, m, N = 12, 4, 50
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
    prob = rat.LQRiskSensitiveProblem(A, B, Q=qs * np.eye(n), R=rw * np.eye(m), N=N, W=1e-3 * np.eye(n), Qf=np.eye(n), kappa=kappa)
    return prob, x0, np.zeros((N, m))


@pytest.mark.parametrize("seed,kappa", [(1, 0.0), (2, 0.0)])
def test_mu_regularisation_restarts_and_iter_max(seed, kappa):
    """Indefinite stage cost: H not PD -> increase_mu_and_delta!, sweep restarts (ileqg.jl:372-378); mu then stays
    above mu_min so the solve runs to iter_max.  Same decisions as the oracle."""
    prob, x0, u = stress_problem(seed, kappa)
    P = orc.Problem(prob)
    ctx = rat.Context(prob, rat.ileqg.make_opts(iter_max=8), max_batch=3, spec_eps=2)
    vg, sg, ig, lg = check_batch(ctx, P, x0, u, np.array([0.0, 1.0, 4.0]), iter_max=8)
    assert np.all(sg == 3) and np.all(ig == 8)
    # operator form reports the regularisation it ended with
    xs = ctx.rollout_open(x0, u)
    ap = ctx.approximate_model(u, xs)
    _, ap_o = orc.approximate_model(P, u, xs)
    st, L, dl, dp, mu, delta = ctx.dp_gain_sweep(ap, 1.0, 0.0, 2.0)
    rc, Lo, dlo, dpo, muo, deo = orc.dp_gain(P, ap_o, 1.0)
    assert st == 0 and rc == 0 and mu == muo and delta == deo and mu > 1e-6
    assert rel(L, Lo) < 1e-9 and rel(dl, dlo) < 1e-9 and rel(dp.s_array, dpo["s"]) < 1e-9


def test_time_varying_cost_and_noise_tables():
    """c(k, x, u) and W(k) depend on k (docs example: c = k/2 x'x + k/2 u'u, optimal_control_problems.jl:50-55)."""
    rng = np.random.default_rng(11)
    n, m, N = 6, 3, 25
    A = 0.95 * np.linalg.qr(rng.standard_normal((n, n)))[0]
    B = rng.standard_normal((n, m)) / np.sqrt(n)
    k = np.arange(N, dtype=float)[:, None, None]
    Q = (0.5 + 0.1 * k) * np.eye(n)
    R = (0.2 + 0.05 * k) * np.eye(m)
    Pm = 0.05 * rng.standard_normal((N, m, n))
    Wk = np.stack([(1e-3 * (1 + 0.5 * np.sin(t))) * np.eye(n) + 1e-4 * np.outer(v, v)
                   for t, v in zip(range(N), rng.standard_normal((N, n)))])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Q, R=R, P=Pm, qv=0.1 * rng.standard_normal((N, n)), rv=0.1 * rng.standard_normal((N, m)),
                                      q0=k.ravel(), N=N, W=Wk, Qf=2 * np.eye(n), qvf=0.3 * rng.standard_normal(n), q0f=1.5)
    x0, u = rng.standard_normal(n), 0.1 * rng.standard_normal((N, m))
    P = orc.Problem(prob)
    ctx = rat.Context(prob, max_batch=6, spec_eps=2)
    check_batch(ctx, P, x0, u, np.array([0.0, 0.5, 2.0, 5.0, 20.0, 200.0]))
    # operator-level parity on the same problem
    _, xo = orc.simulate_open(P, x0, u)
    assert rel(ctx.rollout_open(x0, u), xo) < 1e-12
    ap = ctx.approximate_model(u, xo)
    _, ap_o = orc.approximate_model(P, u, xo)
    a = ap_o.arrays()
    for kk, name in (("q", "q_array"), ("qv", "q_vec_array"), ("Q", "Q_array"), ("r", "r_array"), ("R", "R_array"),
                     ("P", "P_array"), ("A", "A_array"), ("B", "B_array"), ("W", "W_array")):
        assert rel(getattr(ap, name), a[kk]) < 1e-12, kk
    for theta in (0.0, 2.0):
        st, L, dl, dp, _, _ = ctx.dp_gain_sweep(ap, theta, 0.0, 2.0)
        _, Lo, dlo, dpo, _, _ = orc.dp_gain(P, ap_o, theta)
        assert st == 0
        for got, ref in ((L, Lo), (dl, dlo), (dp.s_array, dpo["s"]), (dp.S_array, dpo["S"]), (dp.s_vec_array, dpo["sv"]),
                         (dp.g_array, dpo["g"]), (dp.G_array, dpo["G"]), (dp.H_array, dpo["H"])):
            assert rel(got, ref) < 1e-10


def test_adaptive_eps_init_and_custom_options():
    prob, x0, u = rat.synthetic_lq_problem(seed=11, kappa=0.05)
    P = orc.Problem(prob)
    theta = np.array([0.0, 2.0, 6.0, 8.5])
    for kw in (dict(adaptive_eps_init=True), dict(lam=0.3, eps_init=0.7, d=1e-3), dict(eps_min=0.3, iter_max=3)):
        ctx = rat.Context(prob, rat.ileqg.make_opts(**kw), max_batch=4, spec_eps=3)
        okw = {("adaptive_eps_init" if k == "adaptive_eps_init" else k): (int(v) if k == "adaptive_eps_init" else v) for k, v in kw.items()}
        check_batch(ctx, P, x0, u, theta, **okw)


def test_full_size_batch_properties():
    """BASELINE sizes (B = 1024, E = 8): size-independent properties instead of an oracle run."""
    prob, x0, u = rat.synthetic_lq_problem()
    rng = np.random.default_rng(4)
    theta = np.abs(1.0 + 2.0 * rng.standard_normal(1024))
    theta[::97] = 40.0 + rng.random(theta[::97].size)            # sprinkle infeasible samples
    ctx = rat.Context(prob, max_batch=1024, spec_eps=8)
    v, st, it, ls = ctx.solve_batch(x0, u, theta)
    bad = theta > 12.6
    assert np.all(st[bad] == 1) and np.all(np.isposinf(v[bad])) and np.all(it[bad] == 0)
    ok = theta < 12.5
    assert np.all(st[ok] == 0) and np.all(it[ok] == 2) and np.all(ls[ok] == 2)
    order = np.argsort(theta[ok])
    assert np.all(np.diff(v[ok][order]) >= -1e-9)               # monotone in theta
    # a random subset agrees with the oracle
    idx = rng.choice(1024, 48, replace=False)
    vo, so, _, _ = orc.compute_value_batch(orc.Problem(prob), x0, u, theta[idx], nthreads=8)
    assert np.array_equal(so, st[idx])
    fin = np.isfinite(vo)
    assert np.all(np.abs(v[idx][fin] - vo[fin]) <= VT * np.abs(vo[fin]))
    # E = 1 gives bitwise the same batch
    ctx1 = rat.Context(prob, max_batch=1024, spec_eps=1)
    v1, st1, it1, ls1 = ctx1.solve_batch(x0, u, theta)
    assert np.array_equal(v1, v) and np.array_equal(st1, st) and np.array_equal(ls1, ls)


def test_pruned_speculation_gives_the_same_batch(monkeypatch):
    """Round-based path, E > 1, tile-free candidates: candidate 0's paired wavefronts publish whether the line search will settle on it and the
    evaluations of candidates 1 .. E-1 stop (switch prune).  The sequential rule never reads a candidate behind the one it accepts, so every
    output is the same with the switch off, and the same as E = 1 -- on a workload whose line search backtracks (cubic drift: candidate 0 is
    rejected for part of the samples, which then need the other candidates' values) as on the headline's."""
    for kappa in (0.06, 0.0):
        prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
        B = 1024
        theta = np.where(np.arange(B) % 2 == 0, 2.0, 5.0).astype(float) if kappa else np.abs(1.0 + 2.0 * np.random.default_rng(3).standard_normal(B))
        theta[:3] = [0.0, 30.0, 5.9]
        ref = rat.Context(prob, max_batch=B, spec_eps=1).solve_batch(x0, u, theta)
        if kappa:
            assert (ref[3] > ref[2]).any()                       # some line search really backtracked
        for E in (8, 2):
            ctx = rat.Context(prob, max_batch=B, spec_eps=E)
            assert ctx.get_path(B) == "rounds" and ctx.debug_get("prune") == 1
            got = ctx.solve_batch(x0, u, theta)
            monkeypatch.setenv("RATILQR_PRUNE", "0")
            off = rat.Context(prob, max_batch=B, spec_eps=E)
            monkeypatch.delenv("RATILQR_PRUNE")
            assert off.debug_get("prune") == 0
            plain = off.solve_batch(x0, u, theta)
            for a, b, c in zip(got, plain, ref):
                assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)


def test_spec_eps_is_an_upper_bound(monkeypatch):
    """The DEFAULT policy (no RATILQR_SPEC_FORCE): a handle created with spec_eps = 8 runs the sequential line-search rule on the E = 1
    kernels -- speculation is result-identical and does not pay on this device -- so its batches are the E = 1 handle's
    bit for bit at every batch size, on the headline workload and on one whose line searches backtrack; spec_force = 1 on the live handle
    re-lays it to the requested width (the speculative kernels), same results again."""
    monkeypatch.delenv("RATILQR_SPEC_FORCE", raising=False)
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")            # (bit-identity across paths is a property of the sequential-sweep kernels)
    for kappa in (0.06, 0.0):
        prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
        for B in (1024, 100):
            theta = np.abs(1.0 + 2.0 * np.random.default_rng(3 + B).standard_normal(B)); theta[:3] = [0.0, 30.0, 5.9]
            ref_ctx = rat.Context(prob, max_batch=B, spec_eps=1)
            ref = ref_ctx.solve_batch(x0, u, theta)
            ctx = rat.Context(prob, max_batch=B, spec_eps=8)
            assert ctx.debug_get("spec_width") == 1 and ctx.debug_get("spec_force") == 0 and ctx.get_path(B) == ref_ctx.get_path(B)
            got = ctx.solve_batch(x0, u, theta)
            for a, b in zip(got, ref):
                assert np.array_equal(a, b, equal_nan=True)
            ctx.debug_set("spec_force", 1)                   # the requested width after all: E = 8 kernels on the same handle
            assert ctx.debug_get("spec_width") == 8 and ctx.get_path(B) == ("rounds" if B == 1024 else "block")
            forced = ctx.solve_batch(x0, u, theta)
            for a, b in zip(forced, ref):
                assert np.array_equal(a, b, equal_nan=True)
            ctx.debug_set("spec_force", 0)
            assert ctx.debug_get("spec_width") == 1
            for a, b in zip(ctx.solve_batch(x0, u, theta), ref):
                assert np.array_equal(a, b, equal_nan=True)


def test_api_misuse_is_reported():
    prob, x0, u = rat.synthetic_lq_problem()
    ctx = rat.Context(prob, max_batch=4)
    with pytest.raises(rat.RatError):
        ctx.solve_batch(x0, u, np.ones(5))                       # exceeds max_batch
    big = rat.LQRiskSensitiveProblem(np.eye(33), np.ones((33, 2)), Q=np.eye(33), R=np.eye(2), N=5, W=np.eye(33), Qf=np.eye(33))
    with pytest.raises(rat.RatError):
        rat.Context(big)                                         # n > 32: unsupported, fails loudly (13..32: tests/test_gpu_wide.py)


@pytest.mark.parametrize("seed,theta", [(1, 1.0), (2, 1.0), (2, 4.0)])
def test_dp_failed_line_search_candidates(seed, theta):
    """Indefinite cost + cubic drift: some line-search candidates fail the DP (M not PD / non-finite) and are skipped
    without an eps_history entry or eps_min test (ileqg.jl:529-535, App. B.5); E = 1, 3, 8 replay the same sequence."""
    prob, x0, u = stress_problem(seed, kappa=0.03)
    P = orc.Problem(prob)
    so = orc.ILEQGSolver(P, iter_max=8)
    rc = so.solve(x0, u, theta)
    assert so.s.n_ls_evals > so.s.n_hist            # the oracle did see DP-failed candidates
    ref = None
    for E in (1, 3, 8):
        ctx = rat.Context(prob, rat.ileqg.make_opts(iter_max=8), max_batch=1, spec_eps=E)
        v, st, it, ls = ctx.solve_batch(x0, u, np.array([theta]))
        r = ctx.solve(x0, u, theta)
        assert st[0] == rc and it[0] == so.s.iter_current and ls[0] == so.s.n_ls_evals and r["hist_n"] == so.s.n_hist
        assert np.array_equal(r["eps_history"][:, 0], so.eps_history[:, 0])
        assert abs(v[0] - so.s.value_current) <= 1e-7 * abs(so.s.value_current)
        # E = 1 on a one-sample batch runs the time-parallel sweeps and deviation-form rollouts (solve_block_psw_kernel): equal to rounding;
        # the speculative widths among themselves: bit for bit
        if E == 1:
            v1 = v[0]
            continue
        assert abs(v[0] - v1) <= 1e-9 * abs(v1)
        if ref is None:
            ref = v[0]
        assert v[0] == ref


def test_speculative_gain_sweep_is_result_identical(monkeypatch):
    """Opt-in mode (RATILQR_SPECULATE=1): the next step!'s gain sweep runs on candidate 0 concurrently with its
    evaluation sweep and is committed by the select kernel.  Must not change a single bit of any result."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    theta = np.array([0.0, 1.0, 4.0, 5.0, 5.9, 6.3, 6.6, 9.0])
    sprob, sx0, su = stress_problem(2, kappa=0.03)
    for E in (1, 4):
        ref = rat.Context(prob, max_batch=theta.size, spec_eps=E).solve_batch(x0, u, theta)
        ref_s = rat.Context(sprob, rat.ileqg.make_opts(iter_max=8), max_batch=3, spec_eps=E).solve_batch(sx0, su, np.array([0.0, 1.0, 4.0]))
        monkeypatch.setenv("RATILQR_SPECULATE", "1")
        got = rat.Context(prob, max_batch=theta.size, spec_eps=E).solve_batch(x0, u, theta)
        got_s = rat.Context(sprob, rat.ileqg.make_opts(iter_max=8), max_batch=3, spec_eps=E).solve_batch(sx0, su, np.array([0.0, 1.0, 4.0]))
        sctx = rat.Context(prob, spec_eps=E)
        r1 = sctx.solve(x0, u, 5.0)
        monkeypatch.delenv("RATILQR_SPECULATE")
        r0 = rat.Context(prob, spec_eps=E).solve(x0, u, 5.0)
        for a, b in zip(ref + ref_s, got + got_s):
            assert np.array_equal(a, b)
        assert np.array_equal(r0["L"], r1["L"]) and np.array_equal(r0["x"], r1["x"]) and r0["value"] == r1["value"]
        assert np.array_equal(r0["eps_history"], r1["eps_history"])


def test_dual_sweep_wavefronts_are_result_identical(monkeypatch):
    """Opt-in mode (RATILQR_DUAL=1, E = 1): one wavefront runs candidate 0's policy evaluation and the next step!'s gain sweep
    over a single pass of the tiles.  Same expressions as the separate kernels: every output must be bit-identical."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    theta = np.array([0.0, 1.0, 4.0, 5.0, 5.9, 6.3, 6.6, 9.0, 30.0])
    sprob, sx0, su = stress_problem(2, kappa=0.03)
    lprob, lx0, lu = rat.synthetic_lq_problem()
    th_l = np.concatenate([[0.0], np.linspace(0.01, 14.0, 30), [50.0]])
    ref = rat.Context(prob, max_batch=theta.size).solve_batch(x0, u, theta)
    ref_s = rat.Context(sprob, rat.ileqg.make_opts(iter_max=8), max_batch=3).solve_batch(sx0, su, np.array([0.0, 1.0, 4.0]))
    ref_l = rat.Context(lprob, max_batch=32).solve_batch(lx0, lu, th_l)
    r0 = rat.Context(prob).solve(x0, u, 5.0)
    monkeypatch.setenv("RATILQR_DUAL", "1")
    got = rat.Context(prob, max_batch=theta.size).solve_batch(x0, u, theta)
    got_s = rat.Context(sprob, rat.ileqg.make_opts(iter_max=8), max_batch=3).solve_batch(sx0, su, np.array([0.0, 1.0, 4.0]))
    got_l = rat.Context(lprob, max_batch=32).solve_batch(lx0, lu, th_l)
    r1 = rat.Context(prob).solve(x0, u, 5.0)
    monkeypatch.delenv("RATILQR_DUAL")
    for a, b in zip(ref + ref_s + ref_l, got + got_s + got_l):
        assert np.array_equal(a, b)
    assert np.array_equal(r0["L"], r1["L"]) and np.array_equal(r0["x"], r1["x"]) and r0["value"] == r1["value"]
    assert np.array_equal(r0["eps_history"], r1["eps_history"])


@pytest.mark.parametrize("E", [2, 8])
def test_dual_wavefronts_with_speculative_step_sizes(E, monkeypatch):
    """E > 1 (default; RATILQR_DUAL=0 switches it off): candidate 0 runs in the paired wavefronts (its evaluation + the next gain sweep), candidates 1 .. E-1 in
    the plain evaluation kernel on a second stream.  Bit-identical to the default path, including samples whose accepted candidate
    is not candidate 0 (the speculative gain sweep is then void) and infeasible ones."""
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    theta = np.array([0.0, 1.0, 4.0, 5.0, 5.9, 6.3, 6.6, 9.0, 30.0])
    sprob, sx0, su = stress_problem(2, kappa=0.03)
    pl = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2))
    px0, pu = np.zeros(2), 0.1 * np.ones((10, 2))
    def run():
        out = rat.Context(prob, max_batch=theta.size, spec_eps=E).solve_batch(x0, u, theta)
        out += rat.Context(sprob, rat.ileqg.make_opts(iter_max=8), max_batch=3, spec_eps=E).solve_batch(sx0, su, np.array([0.0, 1.0, 4.0]))
        out += rat.Context(pl, max_batch=3, spec_eps=E).solve_batch(px0, pu, np.array([0.0, 0.5, 2.0]))
        return out
    got = run()                                             # default for E > 1
    monkeypatch.setenv("RATILQR_DUAL", "0")
    ref = run()                                             # every candidate in the plain evaluation kernel, separate gain sweeps
    monkeypatch.delenv("RATILQR_DUAL")
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    assert ref[3].max() > ref[2].max()                      # some line search really backtracked


def test_fused_solve_kernel_equals_round_based_path(monkeypatch):
    """Default for E = 1: one persistent wavefront per theta-sample runs the whole solve! in one launch, pairing each policy
    evaluation with the gain sweep that would follow it (RATILQR_FUSED_DUAL=0: separate passes).  It calls the same device
    functions as the per-phase kernels of the round-based path (RATILQR_FUSED=0): every output must be bit-identical --
    values, statuses (infeasible theta, mu divergence), iteration and line-search counts, trajectories, gains, eps history,
    on the LQ family (with cubic drift, time-varying cost and noise tables via the stress problems) and the power-law family."""
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    theta = np.array([0.0, 1.0, 4.0, 5.0, 5.9, 6.3, 6.6, 9.0, 30.0])
    lprob, lx0, lu = rat.synthetic_lq_problem()
    th_l = np.concatenate([[0.0], np.linspace(0.01, 14.0, 30), [50.0]])
    stress = [stress_problem(i, kappa=0.03) for i in range(4)]
    th_s = np.array([0.0, 0.3, 1.0, 4.0])
    pl = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2), a=1.3, b=1.5, p=2.5, hconst=1.0)
    pl_u = 0.1 * np.ones((10, 2))

    def run_all():
        out = []
        out += rat.Context(prob, max_batch=theta.size).solve_batch(x0, u, theta)
        out += rat.Context(lprob, max_batch=32).solve_batch(lx0, lu, th_l)
        for sp, sx, su in stress:
            out += rat.Context(sp, rat.ileqg.make_opts(iter_max=8), max_batch=4).solve_batch(sx, su, th_s)
        out += rat.Context(pl, max_batch=3).solve_batch(np.zeros(2), pl_u, np.array([0.0, 0.5, 2.0]))
        r = rat.Context(prob).solve(x0, u, 5.0)
        out += [r["L"], r["x"], r["l"], np.array([r["value"]]), np.asarray(r["eps_history"], dtype=float)]
        return out

    monkeypatch.setenv("RATILQR_BLOCK", "0")         # (small batches default to the workgroup-per-sample kernel: tests/test_gpu_block.py)
    fused = run_all()                                # fused, policy evaluation paired with the following gain sweep
    monkeypatch.setenv("RATILQR_FUSED_DUAL", "0")
    fused_plain = run_all()                          # fused, one recursion per pass
    monkeypatch.delenv("RATILQR_FUSED_DUAL")
    monkeypatch.setenv("RATILQR_FUSED_OCC2", "1")
    fused_occ2 = run_all()                           # fused, the 256-register variant that puts two samples on a SIMD (an option since round 3)
    monkeypatch.delenv("RATILQR_FUSED_OCC2")
    monkeypatch.setenv("RATILQR_FUSED", "0")
    rounds = run_all()                               # one launch per phase
    monkeypatch.delenv("RATILQR_FUSED")
    monkeypatch.delenv("RATILQR_BLOCK")
    assert len(fused) == len(rounds) == len(fused_plain) == len(fused_occ2)
    for a, b, c, d in zip(fused, rounds, fused_plain, fused_occ2):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        assert np.array_equal(np.asarray(a), np.asarray(c), equal_nan=True)
        assert np.array_equal(np.asarray(a), np.asarray(d), equal_nan=True)


def test_batches_beyond_one_sample_per_simd_take_the_two_per_simd_kernel(monkeypatch):
    """More samples than the device has SIMDs (LQ family, E = 1): by default the 256-register, tile-free variant of the fused kernel runs two
    samples per SIMD (switch fused_occ2 = -1); the paired kernel run in generations (fused_occ2 = 0) must give the same bits -- also beyond
    the staging horizon, with time-varying cost and noise tables (n = 6, m = 3), with an indefinite stage cost (mu restarts) and cubic drift."""
    import torch
    nsimd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    B = nsimd + 77
    rng = np.random.default_rng(8)
    n, m, N = 6, 3, 25
    k = np.arange(N, dtype=float)[:, None, None]
    Wk = np.stack([(1e-3 * (1 + 0.5 * np.sin(t))) * np.eye(n) + 1e-4 * np.outer(v, v) for t, v in zip(range(N), rng.standard_normal((N, n)))])
    tv = rat.LQRiskSensitiveProblem(0.95 * np.linalg.qr(rng.standard_normal((n, n)))[0], rng.standard_normal((n, m)) / np.sqrt(n),
                                    Q=(0.5 + 0.1 * k) * np.eye(n), R=(0.2 + 0.05 * k) * np.eye(m), P=0.05 * rng.standard_normal((N, m, n)),
                                    qv=0.1 * rng.standard_normal((N, n)), rv=0.1 * rng.standard_normal((N, m)), q0=k.ravel(), N=N, W=Wk,
                                    Qf=2 * np.eye(n), qvf=0.3 * rng.standard_normal(n), q0f=1.5, kappa=0.01)
    cases = [rat.synthetic_lq_problem(seed=5, kappa=0.05), rat.synthetic_lq_problem(n=12, m=4, N=60, seed=4, kappa=0.02), stress_problem(1, kappa=0.03),
             (tv, rng.standard_normal(n), 0.1 * rng.standard_normal((N, m)))]
    for prob, x0, u in cases:
        theta = np.abs(1.0 + 2.0 * rng.standard_normal(B))
        theta[:3] = [0.0, 30.0, 5.9]
        ctx = rat.Context(prob, rat.ileqg.make_opts(iter_max=6), max_batch=B)
        assert ctx.debug_get("fused_occ2") == -1
        got = ctx.solve_batch(x0, u, theta)
        del ctx
        monkeypatch.setenv("RATILQR_FUSED_OCC2", "0")
        ctx = rat.Context(prob, rat.ileqg.make_opts(iter_max=6), max_batch=B)
        ref = ctx.solve_batch(x0, u, theta)
        monkeypatch.delenv("RATILQR_FUSED_OCC2")
        del ctx
        for a, b in zip(got, ref):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("Nh", [1, 3, 7, 13, 60])
def test_horizon_lengths_around_the_unroll_and_staging_limits(Nh):
    """The closed-loop rollout unrolls its time loop by 5 (tail of N mod 5 steps) and, inside the fused solve, stages its operands in
    LDS for N <= 52: horizons on both sides of those limits must agree with the oracle like any other."""
    prob, x0, u = rat.synthetic_lq_problem(n=12, m=4, N=Nh, seed=4, kappa=0.02)
    P = orc.Problem(prob)
    theta = np.array([0.0, 0.7, 2.5])
    ctx = rat.Context(prob, max_batch=theta.size)
    check_batch(ctx, P, x0, u, theta)


def test_batches_larger_than_the_machine():
    """4096 samples on 1024 SIMDs: the fused solve runs its blocks in several waves; values equal the round-based path's."""
    prob, x0, u = rat.synthetic_lq_problem()
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(9).standard_normal(4096))
    v1, s1, i1, l1 = rat.Context(prob, max_batch=4096).solve_batch(x0, u, theta)
    import os
    os.environ["RATILQR_FUSED"] = "0"
    try:
        v0, s0, i0, l0 = rat.Context(prob, max_batch=4096).solve_batch(x0, u, theta)
    finally:
        del os.environ["RATILQR_FUSED"]
    assert np.array_equal(v1, v0) and np.array_equal(s1, s0) and np.array_equal(i1, i0) and np.array_equal(l1, l0)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_and_tables_against_the_oracle(seed, monkeypatch):
    """Fuzz over the shape envelope of the device path (1 <= n <= 12, 1 <= m <= 4, 1 <= N <= 60): random LQ problems with random
    time-varying / time-invariant tables, linear cost terms, cross terms, non-diagonal W and the cubic drift; theta from 0 through
    infeasible; on the default execution path and (odd seeds) on the round-based path with a random speculation width.  Status,
    iteration and line-search counts identical to the oracle's, values to 1e-9."""
    rng = np.random.default_rng(1000 + seed)
    n, m, N = int(rng.integers(1, 13)), int(rng.integers(1, 5)), int(rng.integers(1, 61))
    tv = bool(rng.integers(0, 2))
    A = (0.7 + 0.3 * rng.random()) * np.linalg.qr(rng.standard_normal((n, n)))[0]
    B = rng.standard_normal((n, m)) / np.sqrt(n)
    def spd(k, scale):
        G = rng.standard_normal((k, k))
        return scale * (np.eye(k) + 0.2 * G @ G.T / k)
    if tv:
        Q = np.stack([spd(n, 0.5 + rng.random()) for _ in range(N)])
        R = np.stack([spd(m, 0.1 + 0.3 * rng.random()) for _ in range(N)])
        Pm = 0.03 * rng.standard_normal((N, m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal((N, n)), 0.1 * rng.standard_normal((N, m)), rng.standard_normal(N)
        W = np.stack([spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)])
    else:
        Q, R, Pm = spd(n, 1.0), spd(m, 0.2), 0.03 * rng.standard_normal((m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(m), float(rng.standard_normal())
        W = spd(n, 1e-3)
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Q, R=R, P=Pm, qv=qv, rv=rv, q0=q0, N=N, W=W, Qf=spd(n, 1.0),
                                      qvf=0.2 * rng.standard_normal(n), q0f=float(rng.standard_normal()),
                                      kappa=(0.0 if seed % 3 else 0.02))
    x0, u = rng.standard_normal(n), 0.1 * rng.standard_normal((N, m))
    theta = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-2, 2.5, 9))])
    if seed % 2:
        monkeypatch.setenv("RATILQR_FUSED", "0")
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=(int(rng.integers(1, 6)) if seed % 2 else 1))
    v, st, it, ls = check_batch(ctx, orc.Problem(prob), x0, u, theta)
    assert st[0] == 0                                             # theta = 0 (iLQG) is always feasible


@pytest.mark.parametrize("w", [1e-16, 1e-9, 1.0, 1e2])
def test_noise_scales_from_1e_minus_16_to_1e2(w):
    """W = w I over 18 decades with theta scaled to the same fraction of the breakdown value (theta w = const): the pivot products of
    the elimination and the logdet accumulation neither overflow nor lose the oracle's digits."""
    prob, x0, u = rat.synthetic_lq_problem(w=w)
    theta = np.array([0.0, 1e-4, 1e-3, 5e-3, 1e-2, 2e-2]) / w
    ctx = rat.Context(prob, max_batch=theta.size)
    v, st, it, ls = check_batch(ctx, orc.Problem(prob), x0, u, theta)
    assert st.tolist() == [0, 0, 0, 0, 0, 1]


@pytest.mark.parametrize("seed", range(8))
def test_random_solver_options(seed, monkeypatch):
    """Fuzz over the ILEQGSolver options (lambda, eps_init < 1, eps_min, d, mu_min, Delta_0, iter_max, adaptive eps_init) on cubic-drift
    problems (3 to 15 iterations, iter_max reached, infeasible theta); both execution paths; decisions identical to the oracle's."""
    rng = np.random.default_rng(2000 + seed)
    prob, x0, u = rat.synthetic_lq_problem(seed=int(rng.integers(1, 50)), kappa=float(rng.choice([0.02, 0.04, -0.03])), N=int(rng.integers(8, 41)))
    x0 = x0 * float(rng.uniform(0.4, 1.0))
    kw = dict(lam=float(rng.uniform(0.2, 0.8)), eps_init=float(rng.uniform(0.3, 1.0)), eps_min=float(10.0 ** rng.uniform(-6, -2)),
              d=float(10.0 ** rng.uniform(-4, -1)), mu_min=float(10.0 ** rng.uniform(-8, -4)), Delta_0=float(rng.uniform(1.5, 4.0)),
              iter_max=int(rng.integers(2, 30)), adaptive_eps_init=bool(rng.integers(0, 2)))
    theta = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-1, 1.3, 7))])
    E = 1
    if seed % 2:
        monkeypatch.setenv("RATILQR_FUSED", "0")
        E = int(rng.integers(1, 5))
    ctx = rat.Context(prob, rat.ileqg.make_opts(**kw), max_batch=theta.size, spec_eps=E)
    okw = dict(kw)
    okw["adaptive_eps_init"] = int(okw["adaptive_eps_init"])
    v, st, it, ls = check_batch(ctx, orc.Problem(prob), x0, u, theta, **okw)
    assert (ls >= it).all()


def test_trajectory_that_overflows_fails_like_the_reference():
    """Cubic drift strong enough that the open-loop rollout overflows to Inf / NaN: the reference forms M = inv(W) - theta S and asserts
    isposdef(M) at every theta, 0 included (ileqg.jl:439-440; 0 x Inf = NaN), so initialize! throws -> status 1 and Inf, no hang in the
    regularisation loop.  Both execution paths."""
    prob, x0, u = rat.synthetic_lq_problem(seed=1, kappa=0.1, N=15)
    x0 = 1.4 * x0
    _, xo = orc.simulate_open(orc.Problem(prob), x0, u)
    assert not np.all(np.isfinite(xo))
    theta = np.array([0.0, 0.5, 2.0])
    for E in (1, 3):
        ctx = rat.Context(prob, max_batch=3, spec_eps=E)
        v, st, it, ls = check_batch(ctx, orc.Problem(prob), x0, u, theta)
        assert st.tolist() == [1, 1, 1] and np.all(np.isposinf(v))


def test_negative_and_non_finite_theta():
    """theta < 0 (risk-seeking: M = inv(W) + |theta| S is always positive definite) runs through the same formulas; NaN / +-Inf fail the
    isposdef(M) assert of initialize! like the reference (NaN pivots).  Tile-sized and general-size kernels, every execution path."""
    for n, m, N in ((12, 4, 50), (20, 6, 20)):
        prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=N, kappa=0.02)
        th = np.array([-5.0, -0.5, -0.01, 0.0, 0.3, np.nan, np.inf, -np.inf])
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, th, nthreads=4)
        assert so.tolist() == [0, 0, 0, 0, 0, 1, 1, 1] and np.all(np.diff(vo[:5]) > 0)
        for env in ({}, {"RATILQR_BLOCK": "0"}, {"RATILQR_FUSED": "0"}):
            os.environ.update(env)
            try:
                ctx = rat.Context(prob, max_batch=8)
            finally:
                for k in env:
                    del os.environ[k]
            check_batch(ctx, orc.Problem(prob), x0, u, th)
