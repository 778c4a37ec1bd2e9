"""Pins the oracle's Cross-Entropy loop with the reference's known-answer tests
(/root/reference/test/cross_entropy_bilevel_optimization_test.jl, lines cited; K13-K15) and
checks the control-flow facts of SURVEY.md Appendix B.10-B.14 on injected N(0,1) streams."""
import numpy as np

import ratilqr.jl_amd as rat
from oracle import oracle as orc

N = 10


def nonlinear():                                              # ce_test.jl:14-24
    prob = rat.PowerLawRiskSensitiveProblem(2, N, 0.01 * np.eye(2))
    return orc.Problem(prob), np.zeros(2), 0.1 * np.ones((N, 2))


def test_K13_batch_cost_equals_serial_cost():                 # :27-32
    P, x0, u = nonlinear()
    theta = np.array([0.1, 0.3, 0.43])
    v1, st, _, _ = orc.compute_value_batch(P, x0, u, theta, nthreads=1)
    v2, _, _, _ = orc.compute_value_batch(P, x0, u, theta, nthreads=2)   # "distributed" = 2 workers
    assert np.all(st == 0) and np.all(v1 == v2)
    cost = v1 + 1.0 / theta
    assert np.allclose(cost, [11.002908466254208, 4.33624364124029, 3.3284929065983375], rtol=1e-12)


def test_K14_positive_samples():                              # :34-35
    z = np.random.default_rng(123).standard_normal(200)
    ce = orc.CrossEntropyBilevelOptimizationSolver(z)
    rc, th = ce.get_positive_samples(0.0, 1.0, 10)
    assert rc == 0 and np.all(th > 0) and th.size == 10
    assert np.array_equal(th, z[z > 0][:10])                  # rejection sampling keeps stream order


def test_K15_ce_solve_finite():                               # :37-41
    P, x0, u = nonlinear()
    z = np.random.default_rng(12344).standard_normal(4000)
    ce = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=3)
    rc, th, x, l, L, c_opt, tmin, tmax = ce.solve(P, x0, u, 1.0)
    assert rc == 0 and np.isfinite(c_opt) and not np.isnan(th) and th > 0
    assert tmin <= tmax and ce.c.iter_current == 5


def test_kl_zero_reduces_to_ilqg():                           # ce.jl:386-389,408
    P, x0, u = nonlinear()
    ce = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=3)
    rc, th, x, l, L, val, tmin, tmax = ce.solve(P, x0, u, 0.0)
    assert rc == 0 and th == 0.0 and tmin == 0.0 and tmax == 0.0
    assert np.isclose(val, 1.0029075497782471, rtol=1e-12)


def test_ce_step_bookkeeping_on_lq_problem():                 # App. B.10-B.13
    prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
    P = orc.Problem(prob)
    z = np.random.default_rng(7).standard_normal(10000)
    ce = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=16, num_elite=4, nthreads=2)
    ce.initialize()
    rc, th, cost = ce.step(P, x0, u, 0.1)
    assert rc == 0 and ce.c.iter_current == 1
    valid = np.isfinite(cost)
    # all valid in iteration 1 -> mu_init, sigma_init doubled (B.10); otherwise unchanged or halved
    if valid.all():
        assert ce.c.mu_init == 2.0 and ce.c.sigma_init == 4.0
    order = np.argsort(cost, kind="stable")[:4]
    el = th[order]
    assert np.isclose(ce.c.mu, el.mean(), rtol=1e-14)
    assert np.isclose(ce.c.sigma, np.sqrt(np.mean((el - el.mean()) ** 2)), rtol=1e-12)   # population std (B.13)
    # theta_min/theta_max with the if/elseif quirk (B.12)
    tmin, tmax = np.inf, 0.0
    for t, c in zip(th, cost):
        if np.isinf(c):
            continue
        if t < tmin:
            tmin = t
        elif t > tmax:
            tmax = t
    assert ce.c.theta_min == tmin and ce.c.theta_max == tmax
