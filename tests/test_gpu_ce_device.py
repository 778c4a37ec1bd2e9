"""rat_ce_solve with the Cross-Entropy loop resident on the device (csrc/ce_device.hip: draw and update kernels, one host wait per
solve!) against the SAME call with the switch ce_device = 0 (the host loop: rat_ce_get_positive_samples / rat_ce_update of driver.cpp,
one round trip per CE iteration) -- every field of the solver, theta_opt, the trajectory and the stream position bit for bit -- and
against the oracle (cross_entropy_bilevel_optimization.jl:233-246, 252-335, 364-415)."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
FIELDS = ("mu_init", "sigma_init", "mu", "sigma", "theta_max", "theta_min", "iter_current", "n_solves", "n_redraws", "n_final_retries")


def run(prob, x0, u, rng, device, kl=0.1, n_solves=1, **kw):
    solver = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    ctx = solver.context(prob)
    ctx.debug_set("ce_device", 1 if device else 0)
    outs = []
    for _ in range(n_solves):
        ctx.profile(True)
        ctx.profile_reset()
        out = ce.solve_(solver, prob, x0, u, rng, kl_bound=kl)
        kinds = {k: v["launches"] for k, v in ctx.profile_get().items() if v["launches"]}
        ctx.profile(False)
        outs.append((out, {f: getattr(solver.c, f) for f in FIELDS}, int(rat.native.lib().rat_ce_stream_pos(ctx.h)), kinds))
    return outs


def same(a, b):
    (oa, fa, pa, _), (ob, fb, pb, _) = a, b
    assert fa == fb, (fa, fb)
    assert pa == pb
    assert oa[0] == ob[0] and oa[4] == ob[4] and oa[5] == ob[5] and oa[6] == ob[6]
    for k in (1, 2, 3):
        assert np.array_equal(oa[k], ob[k])


@pytest.mark.parametrize("B,ne,E", [(10, 3, 1), (64, 8, 1), (1024, 100, 1), (100, 10, 2), (37, 5, 8)])
def test_device_loop_equals_host_loop_on_an_injected_stream(B, ne, E):
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(100 + B).standard_normal(40 * B + 4000)
    kw = dict(num_samples=B, num_elite=ne, spec_eps=E)
    dev, host = run(prob, x0, u, z, True, **kw), run(prob, x0, u, z, False, **kw)
    same(dev[0], host[0])
    assert dev[0][3].get("ce_bookkeeping", 0) == 6 and "ce_bookkeeping" not in host[0][3]       # draw | update + draw x 4 | update: one launch between two batches
    assert dev[0][1]["n_solves"] == 5 * B


def test_device_loop_equals_the_oracle_at_full_size():
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(31).standard_normal(30000)
    kw = dict(num_samples=256, num_elite=25)
    (out, f, pos, kinds), = run(prob, x0, u, z, True, **kw)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=8, **kw)
    rc, th, x, l, L, val, tmin, tmax = oc.solve(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0 and kinds.get("ce_bookkeeping") == 6
    assert out[5] == tmin and out[6] == tmax                  # same draws, same valid mask: exact
    assert abs(out[0] - th) <= 1e-9 * th and abs(f["sigma"] - oc.c.sigma) <= 1e-9 * oc.c.sigma and abs(out[4] - val) <= 1e-9 * abs(val)
    assert f["mu_init"] == oc.c.mu_init and f["sigma_init"] == oc.c.sigma_init and f["n_solves"] == oc.c.n_solves


def test_seeded_generator_and_repeated_solves_carry_the_stream_over():
    """The built-in generator: normals generated ahead of the device's need and not consumed go back to the handle's queue, so a second
    solve! on the same solver continues the sequence exactly where the host loop would."""
    prob, x0, u = rat.synthetic_lq_problem()
    kw = dict(num_samples=128, num_elite=12)
    dev, host = run(prob, x0, u, 4242, True, n_solves=3, **kw), run(prob, x0, u, 4242, False, n_solves=3, **kw)
    for a, b in zip(dev, host):
        same(a, b)
    assert dev[1][0][0] != dev[0][0][0]                       # (the second solve really drew new samples)


def test_redraws_inside_the_chain():
    """mu_init far beyond the breakdown: iteration 1 halves (mu_init, sigma_init) and redraws until enough samples are feasible
    (:293-298) -- slots of the device chain are consumed by redraws and the host enqueues the rest; use_theta_max as well."""
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(5).standard_normal(200000)
    for use_max in (False, True):
        kw = dict(num_samples=32, num_elite=4, mu_init=40.0, sigma_init=10.0, use_theta_max=use_max)
        dev, host = run(prob, x0, u, z, True, **kw), run(prob, x0, u, z, False, **kw)
        same(dev[0], host[0])
        assert dev[0][1]["n_redraws"] >= 1
        oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=8, **kw)
        rc, th, *_ = oc.solve(orc.Problem(prob), x0, u, 0.1)
        assert rc == 0 and dev[0][1]["n_redraws"] == oc.c.n_redraws and abs(dev[0][0][0] - th) <= 1e-9 * th


def test_final_solve_retry_and_iter_max_zero():
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(9).standard_normal(5000)
    kw = dict(num_samples=16, num_elite=3, iter_max=0, mu_init=30.0, sigma_init=4.0)      # theta_opt = mu_init is infeasible: lowered by sigma until it solves
    dev, host = run(prob, x0, u, z, True, **kw), run(prob, x0, u, z, False, **kw)
    same(dev[0], host[0])
    assert dev[0][1]["n_final_retries"] >= 1 and dev[0][1]["n_solves"] == 0


def test_stream_exhaustion_is_reported():
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(3).standard_normal(150)          # < 5 x 64 draws
    for device in (True, False):
        solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=64, num_elite=8)
        solver.context(prob).debug_set("ce_device", int(device))
        with pytest.raises(rat.native.RatError, match="STREAM_DRY"):
            ce.solve_(solver, prob, x0, u, z, kl_bound=0.1)


def test_a_short_upload_is_topped_up():
    """sigma >> mu: half the draws are rejected and the normals provisioned per slot run out on the device; the chain reports it, the host
    uploads more and the solve continues -- same result as the host loop."""
    prob, x0, u = rat.synthetic_lq_problem()
    z = np.random.default_rng(77).standard_normal(120000)
    kw = dict(num_samples=512, num_elite=50, mu_init=-2.0, sigma_init=1.0)               # P(theta > 0) ~ 2 %: a draw needs ~ 22,000 normals, a slot provisions 1,088
    dev, host = run(prob, x0, u, z, True, **kw), run(prob, x0, u, z, False, **kw)
    same(dev[0], host[0])


def test_update_kernel_equals_the_host_update_on_ties_inf_nan_and_signed_zeros():
    """rat_ce_update_dev (the update kernel of the device-resident loop on injected costs) against rat_ce_update (host) and the oracle's
    elite update: every field bit for bit on costs with exact ties, +-Inf, NaN and both signed zeros -- sort(by = cost) orders by isless,
    and isless(-0.0, 0.0) is true (cross_entropy...jl:326-328)."""
    import ctypes as C
    nv = rat.native
    L = nv.lib()
    prob, x0, u = rat.synthetic_lq_problem()
    ctx = rat.Context(prob, max_batch=16)
    rng = np.random.default_rng(7)
    cases = [(np.array([5.0, 1.0, 7.0, 2.0]), np.array([0.0, -0.0, 0.0, -0.0]), k) for k in (1, 2, 3)]
    for trial in range(60):
        B = int(rng.integers(4, 1025 if trial % 10 == 0 else 80))
        k = int(rng.integers(1, B // 2 + 1))
        theta = rng.uniform(0.1, 5.0, B)
        cost = rng.integers(-2, 3, B).astype(float)
        cost[rng.random(B) < 0.3] *= -0.0                                   # signed zeros of both kinds (x * -0.0 = -+0.0)
        cost[rng.random(B) < 0.10] = np.inf
        cost[rng.random(B) < 0.03] = -np.inf
        if trial % 3 == 0:
            cost[rng.random(B) < 0.1] = np.nan
        cases.append((theta, cost, k))
    for theta, cost, k in cases:
        B = theta.size
        for it in (1, 2):
            cs = []
            for dev in (False, True):
                c = nv.CeSolver()
                L.rat_ce_default(C.byref(c))
                c.num_samples, c.num_elite, c.iter_current, c.lam = B, k, it, 0.5
                c.theta_min, c.theta_max = 1.5, 2.5
                redraw = C.c_int32()
                if dev:
                    nv.check(L.rat_ce_update_dev(ctx.h, C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
                else:
                    nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
                cs.append((redraw.value, c.mu, c.sigma, c.mu_init, c.sigma_init, c.theta_min, c.theta_max))
            assert cs[0] == cs[1], (B, k, it, cs)
            if cs[0][0] == 0:
                oc = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=B, num_elite=k)
                oc.c.theta_min, oc.c.theta_max = 1.5, 2.5
                oc.elite_update(theta, cost)
                assert (oc.c.mu, oc.c.sigma, oc.c.theta_min, oc.c.theta_max) == (cs[0][1], cs[0][2], cs[0][5], cs[0][6])
