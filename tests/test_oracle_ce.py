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


# ---- the two solve! branches the reference's own tests never reach (VERDICT r01 missing #5) --------------------------------------
def quirk_stream(thetas, mu_init=1.0, sigma_init=2.0):
    """N(0,1) draws that make get_positive_samples (:233-246) return exactly `thetas` in iteration 1."""
    return (np.asarray(thetas, float) - mu_init) / sigma_init


def test_use_theta_max_returns_theta_max_with_the_if_elseif_quirk():          # :318-322, :375-379
    """theta_min / theta_max are updated by `if theta < theta_min ... elseif theta > theta_max`: a sample that lowers theta_min cannot
    raise theta_max in the same pass, so the FIRST valid sample never counts towards theta_max.  With the first sample also the largest
    one, use_theta_max = true therefore returns the second-largest theta -- what the reference does, not what its docstring says."""
    P, x0, u = nonlinear()
    thetas = [0.45, 0.1, 0.3]                                  # first valid sample is the largest
    drawn = 1.0 + 2.0 * quirk_stream(thetas)                   # what rand(Normal(1, 2)) returns for this stream (last-bit rounding)
    ce = orc.CrossEntropyBilevelOptimizationSolver(quirk_stream(thetas), num_samples=3, iter_max=1, use_theta_max=True)
    rc, th, x, l, L, val, tmin, tmax = ce.solve(P, x0, u, 1.0)
    assert rc == 0 and tmin == drawn[1] and tmax == drawn[2] and th == drawn[2]          # not 0.45
    assert np.isclose(val, 4.33624364124029, rtol=1e-9)                    # value(0.3) + 1/0.3 (K13 anchor)
    assert ce.c.n_final_retries == 0 and ce.c.mu_init == 2.0 and ce.c.sigma_init == 4.0   # all valid in iteration 1 (:299-305)
    # same samples, largest one last: now it is seen
    ce2 = orc.CrossEntropyBilevelOptimizationSolver(quirk_stream([0.1, 0.3, 0.45]), num_samples=3, iter_max=1, use_theta_max=True)
    rc, th2, *_rest, tmin2, tmax2 = ce2.solve(P, x0, u, 1.0)
    d2 = 1.0 + 2.0 * quirk_stream([0.1, 0.3, 0.45])
    assert rc == 0 and th2 == d2[2] and tmin2 == d2[0] and tmax2 == d2[2]
    # use_theta_max = false on the first stream: theta_opt = mu of the elites (all three samples here)
    ce3 = orc.CrossEntropyBilevelOptimizationSolver(quirk_stream(thetas), num_samples=3, iter_max=1)
    rc, th3, *_ = ce3.solve(P, x0, u, 1.0)
    assert rc == 0 and np.isclose(th3, np.mean(drawn), rtol=1e-15)


def test_final_solve_retry_lowers_theta_opt_by_sigma():                       # :410-413
    """iter_max = 0 skips the CE loop: theta_opt = mu = mu_init (reset by initialize!, :133-138) and sigma = sigma_init.  With mu_init above
    the breakdown the final solve throws and theta_opt <- max(0, theta_opt - sigma) until it is feasible."""
    prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
    P = orc.Problem(prob)
    v, st, _, _ = orc.compute_value_batch(P, x0, u, np.array([40.0, 30.0, 20.0, 10.0, 0.0]), nthreads=1)
    fails = [int(s != 0) for s in st]
    assert fails[0] == 1 and fails[-1] == 0
    want_retries = fails.index(0)
    ce = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=3, iter_max=0, mu_init=40.0, sigma_init=10.0)
    rc, th, x, l, L, val, tmin, tmax = ce.solve(P, x0, u, 0.1)
    assert rc == 0 and ce.c.n_final_retries == want_retries >= 1 and th == 40.0 - 10.0 * want_retries
    assert np.isinf(tmin) and tmax == 0.0                                   # untouched by a CE loop that never ran
    assert val == v[want_retries] + (0.1 / th if th > 0 else np.inf) or (th == 0.0 and np.isinf(val))
    # clamp at zero: sigma larger than mu
    ce = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=3, iter_max=0, mu_init=40.0, sigma_init=100.0)
    rc, th, *_ = ce.solve(P, x0, u, 0.1)
    assert rc == 0 and th == 0.0 and ce.c.n_final_retries == 1
