"""GPU restatement of /root/reference/test/cross_entropy_bilevel_optimization_test.jl (K13-K15) and CE parity
against the oracle on injected N(0,1) streams (identical valid mask / elite set, mu and sigma to 1e-9)."""
import json
import os

import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
N = 10


def nonlinear():                                              # ce_test.jl:14-24
    prob = rat.PowerLawRiskSensitiveProblem(2, N, 0.01 * np.eye(2))
    return prob, np.zeros(2), 0.1 * np.ones((N, 2))


def test_compute_cost_equals_serial():                        # :27-32
    prob, x0, u = nonlinear()
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    ce.initialize_(solver)
    theta = np.array([0.1, 0.3, 0.43])
    costs = rat.compute_cost(solver, prob, x0, u, theta, 1.0)
    costs_test = rat.compute_cost_serial(solver, prob, x0, u, theta, 1.0)
    assert np.allclose(costs, costs_test, rtol=1.5e-8, atol=0)
    assert np.array_equal(costs, costs_test)                  # batch composition must not matter at all
    assert np.allclose(costs, [11.002908466254208, 4.33624364124029, 3.3284929065983375], rtol=1e-9)


def test_compute_value_worker_is_value_or_inf():              # ce.jl:144-167
    prob, x0, u = nonlinear()
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    costs = ce.compute_cost(solver, prob, x0, u, [0.1, 0.3, 0.43], 1.0)
    vals = [rat.compute_value_worker(solver, prob, x0, u, th) for th in (0.1, 0.3, 0.43)]
    assert np.array_equal(np.array(vals) + 1.0 / np.array([0.1, 0.3, 0.43]), costs)
    lq, lx0, lu = rat.synthetic_lq_problem()
    assert np.isposinf(rat.compute_value_worker(rat.CrossEntropyBilevelOptimizationSolver(), lq, lx0, lu, 60.0))   # exception -> Inf (:163-165)


def test_get_positive_samples():                              # :34-35
    z = np.random.default_rng(123).standard_normal(200)
    th = rat.get_positive_samples(0.0, 1.0, 10, z)
    assert np.all(th > 0) and th.size == 10
    prob, x0, u = nonlinear()
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    th2 = rat.get_positive_samples(0.0, 1.0, 1000, 123, ce_solver=solver, problem=prob)   # built-in generator
    assert np.all(th2 > 0) and th2.size == 1000
    assert abs(th2.mean() - np.sqrt(2 / np.pi)) < 0.08        # half-normal mean


def test_ce_solve_smoke():                                    # :37-41
    prob, x0, u = nonlinear()
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    th_opt, x, l, L, c_opt, tmin, tmax = ce.solve_(solver, prob, x0, u, 12344, kl_bound=1.0)
    assert np.isfinite(c_opt) and not np.isnan(th_opt)


def test_ce_solve_matches_oracle_on_injected_stream():
    prob, x0, u = nonlinear()
    z = np.random.default_rng(12344).standard_normal(4000)
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    got = ce.solve_(solver, prob, x0, u, z, kl_bound=1.0)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=3)
    rc, th, x, l, L, val, tmin, tmax = oc.solve(orc.Problem(prob), x0, u, 1.0)
    assert rc == 0
    assert abs(got[0] - th) <= 1e-9 * abs(th) and abs(got[4] - val) <= 1e-9 * abs(val)
    assert got[5] == tmin and got[6] == tmax
    assert solver.c.mu_init == oc.c.mu_init and solver.c.sigma_init == oc.c.sigma_init
    assert np.abs(got[1] - x).max() < 1e-9 and np.abs(got[3] - L).max() < 1e-9


def test_ce_config_golden():
    """CE on the N=50, n=12, m=4 LQ problem: 64 samples, 8 elites, stream seed 2024 (tests/golden)."""
    g = GOLD["ce_config"]
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(g["z_seed"]).standard_normal(20000)
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=g["num_samples"], num_elite=g["num_elite"], spec_eps=2)
    th, x, l, L, val, tmin, tmax = ce.solve_(solver, prob, x0, u, z, kl_bound=g["kl_bound"])
    assert abs(th - g["theta_opt"]) <= 1e-9 * g["theta_opt"] and abs(val - g["value"]) <= 1e-9 * g["value"]
    assert tmin == g["theta_min"] and tmax == g["theta_max"]
    assert abs(solver.c.mu - g["mu"]) <= 1e-9 * g["mu"] and abs(solver.c.sigma - g["sigma"]) <= 1e-6 * g["sigma"] + 1e-12
    assert solver.c.mu_init == g["mu_init"] and solver.c.sigma_init == g["sigma_init"] and solver.c.n_solves == g["n_solves"]
    assert np.abs(l[0] - np.array(g["l0"])).max() < 1e-9 and np.abs(L[0] - np.array(g["L0"])).max() < 1e-9
    assert rat.native.lib().rat_ce_stream_pos(solver.context(prob).h) == g["zpos"]


def test_ce_step_redraw_when_too_few_valid():
    """Iteration 1 with mostly infeasible samples: mu_init, sigma_init halve and the batch is redrawn (:293-298)."""
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(5).standard_normal(50000)
    kw = dict(num_samples=32, num_elite=4, mu_init=40.0, sigma_init=10.0)
    solver = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    ce.initialize_(solver)
    th, cost = ce.step_(solver, prob, x0, u, 0.1, z)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=8, **kw)
    oc.initialize()
    rc, tho, costo = oc.step(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0 and solver.c.n_redraws == oc.c.n_redraws and solver.c.n_redraws >= 1
    assert solver.c.mu_init == oc.c.mu_init and solver.c.sigma_init == oc.c.sigma_init
    assert np.array_equal(th, tho) and np.array_equal(np.isinf(cost), np.isinf(costo))
    assert abs(solver.c.mu - oc.c.mu) <= 1e-9 * oc.c.mu and abs(solver.c.sigma - oc.c.sigma) <= 1e-9 * oc.c.sigma
    assert solver.c.theta_min == oc.c.theta_min and solver.c.theta_max == oc.c.theta_max


def test_kl_zero_is_ilqg():                                   # ce.jl:386-389, 408
    prob, x0, u = nonlinear()
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3)
    th, x, l, L, val, tmin, tmax = ce.solve_(solver, prob, x0, u, 1, kl_bound=0.0)
    assert th == 0.0 and tmin == 0.0 and tmax == 0.0 and abs(val - 1.0029075497782471) < 1e-9


def test_config3_full_size_ce_matches_the_oracle(monkeypatch):
    """BASELINE config 3: the full RAT iLQR solve -- 1024 CE samples, 100 elites, 5 CE iterations, N = 50, n = 12, m = 4 -- with one
    (fused solve kernel) and with 8 speculative line-search step sizes per sample, against the oracle run live on the same injected
    N(0,1) stream (5120 iLEQG solves on the host cores): same draws, same elite sets (hence mu, sigma to 1e-9), same theta_opt."""
    import os
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(31).standard_normal(40000)
    kw = dict(num_samples=1024, num_elite=100)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=os.cpu_count() or 8, **kw)
    rc, th_o, x_o, l_o, L_o, val_o, tmin_o, tmax_o = oc.solve(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0
    res = []
    for E, force in ((1, True), (8, True), (8, False)):             # (8, False): the default policy -- spec_eps as an upper bound, sequential rule
        if force:
            monkeypatch.setenv("RATILQR_SPEC_FORCE", "1")
        else:
            monkeypatch.delenv("RATILQR_SPEC_FORCE", raising=False)
        solver = rat.CrossEntropyBilevelOptimizationSolver(spec_eps=E, **kw)
        th, x, l, L, val, tmin, tmax = ce.solve_(solver, prob, x0, u, z, kl_bound=0.1)
        assert abs(th - th_o) <= 1e-9 * th_o and abs(val - val_o) <= 1e-9 * abs(val_o)
        assert tmin == tmin_o and tmax == tmax_o                                   # extreme samples: identical draws
        assert abs(solver.c.mu - oc.c.mu) <= 1e-9 * oc.c.mu and abs(solver.c.sigma - oc.c.sigma) <= 1e-6 * oc.c.sigma + 1e-12
        assert solver.c.mu_init == oc.c.mu_init and solver.c.sigma_init == oc.c.sigma_init
        assert solver.c.n_solves == 5 * 1024
        assert np.abs(x - x_o).max() < 1e-9 and np.abs(L - L_o).max() < 1e-9
        assert solver.context(prob).debug_get("spec_width") == (E if force else 1)
        res.append((th, val, solver.c.mu, solver.c.sigma, x, l, L))
    # speculation width does not change a bit of the batches (mu, sigma, theta_opt); the final solve at theta_opt is ONE sample, which
    # the E = 1 handle runs with time-parallel sweeps (solve_block_psw_kernel): its outputs agree to rounding
    for other in (res[1], res[2]):
        for k, (a, b) in enumerate(zip(res[0], other)):
            if k in (0, 2, 3):
                assert np.array_equal(np.asarray(a), np.asarray(b))
            else:
                assert np.allclose(np.asarray(a), np.asarray(b), rtol=1e-12, atol=1e-13)


# ---- the two solve! branches the reference's own tests never reach (VERDICT r01 missing #5), GPU vs oracle ------------------------
def test_use_theta_max_with_the_if_elseif_quirk_matches_the_oracle():        # :318-322, :375-379
    prob, x0, u = nonlinear()
    for thetas, iw in (([0.45, 0.1, 0.3], 2), ([0.1, 0.3, 0.45], 2)):
        z = (np.array(thetas) - 1.0) / 2.0
        want, low = (1.0 + 2.0 * z)[iw], (1.0 + 2.0 * z).min()
        solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3, iter_max=1, use_theta_max=True)
        got = ce.solve_(solver, prob, x0, u, z, kl_bound=1.0)
        oc = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=3, iter_max=1, use_theta_max=True)
        rc, th, x, l, L, val, tmin, tmax = oc.solve(orc.Problem(prob), x0, u, 1.0)
        assert rc == 0 and got[0] == th == want and got[5] == tmin == low and got[6] == tmax == want
        assert abs(got[4] - val) <= 1e-9 * abs(val) and np.abs(got[3] - L).max() < 1e-9
        assert solver.c.n_final_retries == oc.c.n_final_retries == 0


def test_final_solve_retry_matches_the_oracle():                              # :410-413
    prob, x0, u = rat.synthetic_lq_problem()
    for mu0, s0 in ((40.0, 10.0), (40.0, 100.0), (25.0, 6.0)):
        solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=3, iter_max=0, mu_init=mu0, sigma_init=s0)
        got = ce.solve_(solver, prob, x0, u, np.zeros(1), kl_bound=0.1)
        oc = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=3, iter_max=0, mu_init=mu0, sigma_init=s0)
        rc, th, x, l, L, val, tmin, tmax = oc.solve(orc.Problem(prob), x0, u, 0.1)
        assert rc == 0 and got[0] == th and solver.c.n_final_retries == oc.c.n_final_retries >= 1
        assert (np.isinf(val) and np.isinf(got[4])) or abs(got[4] - val) <= 1e-9 * abs(val)
        assert np.isinf(got[5]) and got[6] == 0.0 and np.abs(got[1] - x).max() < 1e-9


def test_builtin_generator_sequence_is_unchanged_by_drawing_ahead():
    """rat_ce_step draws the NEXT iteration's normals on the host while its batch runs on the device (same xoshiro / Box-Muller sequence,
    only earlier); the thetas it hands out must be what plain get_positive_samples calls draw from the same seed, rejections included."""
    prob, x0, u = rat.synthetic_lq_problem()
    kw = dict(num_samples=48, num_elite=6, mu_init=0.5, sigma_init=2.0)          # mu_init / sigma_init: a good share of rejected draws
    a = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    ce.initialize_(a)
    got, params = [], []
    for _ in range(3):
        params.append((a.c.mu_init, a.c.sigma_init) if a.c.iter_current == 0 else (a.c.mu, a.c.sigma))
        th, _ = ce.step_(a, prob, x0, u, 0.1, 777)
        assert a.c.n_redraws == 0
        got.append(th)
    b = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    for (mu, sigma), th in zip(params, got):
        assert np.array_equal(rat.get_positive_samples(mu, sigma, 48, 777, ce_solver=b, problem=prob), th)
    pos_a = nv_pos(a, prob)
    assert pos_a == nv_pos(b, prob) and pos_a > 3 * 48                                # same number of normals consumed, some rejected


def nv_pos(solver, prob):
    from ratilqr.jl_amd import _native as nv
    return int(nv.lib().rat_ce_stream_pos(solver.context(prob).h))
