"""CPU-side checks of the product library: it loads without a GPU, exports exactly what include/ratilqr.h
declares, validates options like the reference's @asserts, refuses to compute without a device (no CPU
fallback), and its host-side CE bookkeeping (rat_ce_draw_stream / rat_ce_update) replays the oracle's
step! decisions when fed the same costs."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import _native as nv
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_GPU = torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ratilqr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rat_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = nv.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ratilqr.h but not exported"
    assert declared == set(nv.EXPORTS)
    assert lib.rat_version() == 600


def test_defaults_match_reference_constructor():                    # ileqg.jl:191-194, ce.jl:100-116
    o = nv.IleqgOpts()
    nv.lib().rat_default_ileqg_opts(C.byref(o))
    assert (o.mu_min, o.delta_0, o.lam, o.d, o.iter_max, o.eps_init, o.eps_min, o.adaptive_eps_init) == \
           (1e-6, 2.0, 0.5, 1e-2, 100, 1.0, 1e-6, 0)
    c = nv.CeSolver()
    nv.lib().rat_ce_default(C.byref(c))
    assert (c.num_samples, c.num_elite, c.iter_max, c.lam, c.mu_init, c.sigma_init) == (10, 3, 5, 0.5, 1.0, 2.0)
    assert c.theta_max == 0.0 and np.isinf(c.theta_min)


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    prob, x0, u = rat.synthetic_lq_problem()
    with pytest.raises(rat.RatError):
        rat.Context(prob)


def test_option_asserts_on_host():                                  # ileqg.jl:195-201
    for bad in (dict(lam=0.0), dict(d=-1.0), dict(mu_min=0.0), dict(Delta_0=-2.0), dict(eps_init=0.0),
                dict(eps_init=0.5, eps_min=0.6), dict(eps_min=0.0)):
        with pytest.raises(AssertionError):
            rat.ileqg.make_opts(**bad)


def test_problem_tables_are_column_major():
    prob, _, _ = rat.synthetic_lq_problem(n=3, m=2, N=4, seed=2)
    t = prob.c_tables()
    assert np.array_equal(t["A"].reshape(3, 3).T, prob.A) and np.array_equal(t["B"].reshape(2, 3).T, prob.B)
    d, keep = nv.make_desc(prob)
    assert (d.n, d.m, d.N, d.model) == (3, 2, 4, 1)


def test_ce_host_bookkeeping_replays_oracle_step():
    """rat_ce_begin_step / rat_ce_draw_stream / rat_ce_update (pure host code in the .so) against orc_ce_step."""
    L = nv.lib()
    prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
    P = orc.Problem(prob)
    z = np.random.default_rng(3).standard_normal(20000)
    for kw in (dict(num_samples=16, num_elite=4), dict(num_samples=16, num_elite=4, mu_init=30.0, sigma_init=8.0)):
        oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=4, **kw)
        oc.initialize()
        c = nv.CeSolver()
        L.rat_ce_default(C.byref(c))
        c.num_samples, c.num_elite = kw["num_samples"], kw["num_elite"]
        c.mu_init, c.sigma_init = kw.get("mu_init", 1.0), kw.get("sigma_init", 2.0)
        L.rat_ce_initialize(C.byref(c))
        zpos = C.c_int64(0)
        for it in range(3):
            rc, tho, costo = oc.step(P, x0, u, 0.1)
            assert rc == 0
            nv.check(L.rat_ce_begin_step(C.byref(c)))
            theta = np.zeros(c.num_samples)
            while True:
                nv.check(L.rat_ce_draw_stream(C.byref(c), nv.P(z), C.c_int64(z.size), C.byref(zpos), nv.P(theta)))
                val, _, _, _ = orc.compute_value_batch(P, x0, u, theta, nthreads=4)
                cost = val + 0.1 / theta
                redraw = C.c_int32()
                nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
                if not redraw.value:
                    break
            assert np.array_equal(theta, tho) and zpos.value == oc.c.zpos
            assert c.mu == oc.c.mu and c.sigma == oc.c.sigma
            assert c.mu_init == oc.c.mu_init and c.sigma_init == oc.c.sigma_init
            assert c.theta_min == oc.c.theta_min and c.theta_max == oc.c.theta_max and c.iter_current == oc.c.iter_current


def test_ce_elite_selection_equals_a_stable_sort_with_ties_inf_and_nan():
    """rat_ce_update picks the elites by a partial sort under (isless(cost), input index): exactly the first num_elite entries of the
    reference's stable sort(by = cost) (cross_entropy...jl:326-330), whatever the ties, +Inf and NaN costs."""
    L = nv.lib()
    rng = np.random.default_rng(0)
    for trial in range(200):
        B = int(rng.integers(4, 40))
        k = int(rng.integers(1, B // 2 + 1))
        theta = rng.uniform(0.1, 5.0, B)
        cost = rng.integers(0, 4, B).astype(float)                       # many exact ties
        cost[rng.random(B) < 0.15] = np.inf
        if trial % 3 == 0:
            cost[rng.random(B) < 0.1] = np.nan
        n_valid = int(np.sum(~np.isinf(cost)))
        c = nv.CeSolver()
        L.rat_ce_default(C.byref(c))
        c.num_samples, c.num_elite, c.iter_current, c.lam = B, k, 2, 0.0      # (iteration 2, lambda 0: accepted whenever num_valid >= num_elite)
        redraw = C.c_int32()
        nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
        if n_valid < k:
            assert redraw.value == 1
            continue
        assert redraw.value == 0
        key = np.where(np.isnan(cost), np.inf, cost)                          # isless: NaN after everything, +Inf included
        order = sorted(range(B), key=lambda i: (np.isnan(cost[i]), key[i]))   # Python's sort is stable
        el = theta[order[:k]]
        mu = sum(el.tolist()) / k                                             # (left-to-right, as the library and the reference sum)
        assert c.mu == mu, (trial, c.mu, mu)
        assert c.sigma == float(np.sqrt(sum(((t - mu) * (t - mu)) for t in el.tolist()) / k))


def test_elite_order_is_julias_isless_on_signed_zeros():
    """sort(by = cost) orders by Base.isless, and isless(-0.0, 0.0) is TRUE (cross_entropy...jl:326-328): with costs
    [0.0, -0.0, 0.0, -0.0] the stable sort puts samples 1, 3 before 0, 2.  Host update and oracle; the device update kernel is held to the
    same in tests/test_gpu_ce_device.py."""
    L = nv.lib()
    lo = orc.lib()
    lo.orc_isless.argtypes = [C.c_double, C.c_double]
    assert lo.orc_isless(-0.0, 0.0) == 1 and lo.orc_isless(0.0, -0.0) == 0 and lo.orc_isless(0.0, 0.0) == 0
    assert lo.orc_isless(1.0, float("nan")) == 1 and lo.orc_isless(float("nan"), float("inf")) == 0 and lo.orc_isless(-1.0, -0.0) == 1
    theta = np.array([5.0, 1.0, 7.0, 2.0])
    cost = np.array([0.0, -0.0, 0.0, -0.0])
    for k, want in ((1, [1.0]), (2, [1.0, 2.0]), (3, [1.0, 2.0, 5.0])):
        c = nv.CeSolver()
        L.rat_ce_default(C.byref(c))
        c.num_samples, c.num_elite, c.iter_current, c.lam = 4, k, 2, 0.0
        redraw = C.c_int32()
        nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
        mu = sum(want) / k
        assert redraw.value == 0 and c.mu == mu and c.sigma == float(np.sqrt(sum((t - mu) * (t - mu) for t in want) / k))
        oc = orc.CrossEntropyBilevelOptimizationSolver(np.zeros(1), num_samples=4, num_elite=k)
        oc.elite_update(theta, cost)
        assert oc.c.mu == c.mu and oc.c.sigma == c.sigma


def test_stream_exhaustion_is_an_error():
    L = nv.lib()
    c = nv.CeSolver()
    L.rat_ce_default(C.byref(c))
    L.rat_ce_begin_step(C.byref(c))
    z = -np.ones(50)
    th = np.zeros(10)
    zp = C.c_int64(0)
    rc = L.rat_ce_draw_stream(C.byref(c), nv.P(z), C.c_int64(z.size), C.byref(zp), nv.P(th))
    assert rc == 5 and b"exhausted" in L.rat_last_error()


def test_ce_update_has_the_if_elseif_theta_max_quirk():
    """rat_ce_update (device-free): the first valid sample only sets theta_min (cross_entropy...jl:318-322), as the oracle does."""
    L = nv.lib()
    c = nv.CeSolver()
    L.rat_ce_default(C.byref(c))
    c.num_samples, c.num_elite = 3, 3
    L.rat_ce_initialize(C.byref(c))
    nv.check(L.rat_ce_begin_step(C.byref(c)))
    theta, cost = np.array([0.45, 0.1, 0.3]), np.array([3.0, 11.0, 4.3])
    redraw = C.c_int32()
    nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
    assert redraw.value == 0 and c.theta_min == 0.1 and c.theta_max == 0.3
    assert (c.mu_init, c.sigma_init) == (2.0, 4.0) and abs(c.mu - theta.mean()) < 1e-15
    # an infeasible first sample is skipped: the next valid one takes its place as "first"
    L.rat_ce_initialize(C.byref(c))
    c.num_elite = 2                                            # 2 valid of 3 >= max(num_elite, num_samples * lambda): no redraw (:306)
    nv.check(L.rat_ce_begin_step(C.byref(c)))
    theta, cost = np.array([9.0, 0.45, 0.3]), np.array([np.inf, 3.0, 4.3])
    nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(cost), C.byref(redraw)))
    assert redraw.value == 0 and c.theta_min == 0.3 and c.theta_max == 0.0 and (c.mu_init, c.sigma_init) == (2.0, 4.0)


def test_shard_bounds_in_c_match_the_python_layer():
    """rat_shard_bounds (the block arithmetic rat_multi_* uses on the device side) == ratilqr.jl_amd.distributed.shard_bounds (what the
    gloo world-size-2/3 tests exercise): contiguous, exhaustive, blocks differ by at most one sample."""
    from ratilqr.jl_amd import distributed as rd
    from ratilqr.jl_amd import multi
    for B in (0, 1, 5, 7, 1000, 1024, 10000):
        for world in (1, 2, 3, 8):
            blocks = [multi.shard_bounds(B, world, r) for r in range(world)]
            assert blocks == [rd.shard_bounds(B, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    L = nv.lib()
    lo, hi = C.c_int64(), C.c_int64()
    assert L.rat_shard_bounds(C.c_int64(10), C.c_int32(4), C.c_int32(4), C.byref(lo), C.byref(hi)) == 1      # rank out of range


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_create_multi_without_devices_fails_loudly():
    m = C.c_void_p()
    rc = nv.lib().rat_create_multi(None, 64, 1, 2, None, C.byref(m))
    assert rc != 0 and not m.value


def test_pets_host_bookkeeping_replays_the_oracle():
    """rat_pets_sample_controls / rat_pets_update (device-free host code of the .so, pets.jl:159-191, 206-216) against the oracle's
    update on the same costs: elites, smoothed means, diagonal variances."""
    from ratilqr.jl_amd import pets
    L = nv.lib()
    rng = np.random.default_rng(8)
    Nh, m, S = 6, 3, 20
    mu0 = rng.standard_normal((Nh, m))
    Sig0 = np.stack([np.diag(rng.uniform(0.5, 2.0, m)) + 0.1 * np.ones((m, m)) for _ in range(Nh)])
    ds = pets.CrossEntropyDirectOptimizationSolver(mu0, Sig0, num_control_samples=S, num_trajectory_samples=4, num_elite=5, smoothing_factor=0.3)
    zc = rng.standard_normal((S, Nh, m))
    ctrl = np.zeros((S, Nh, m))
    nv.check(L.rat_pets_sample_controls(C.byref(ds.c), nv.P(nv.f64(zc)), nv.P(ctrl)))
    want = mu0[None] + np.einsum("tab,stb->sta", np.linalg.cholesky(Sig0), zc)            # rand(rng, MvNormal(mu_t, Sigma_t))
    assert np.allclose(ctrl, want, rtol=0, atol=1e-13)
    cost = rng.standard_normal(S)
    cost[3] = np.nan                                                                        # sorts last (isless)
    idx = np.zeros(5, np.int64)
    nv.check(L.rat_pets_update(C.byref(ds.c), nv.P(ctrl), nv.P(cost), idx.ctypes.data_as(C.POINTER(C.c_int64))))
    op = orc.PetsSolver(mu0, Sig0, num_control_samples=S, num_trajectory_samples=4, num_elite=5, smoothing_factor=0.3)
    idx_o = op.update(ctrl, cost)
    assert np.array_equal(idx, idx_o) and 3 not in idx
    assert np.allclose(ds.mu_array, op.mu_array, rtol=1e-15, atol=0) and np.allclose(ds.Sigma_array, op.Sigma_array, rtol=1e-14, atol=1e-16)
    bad = Sig0.copy()
    bad[2] = -np.eye(m)                                                                     # MvNormal would throw
    ds2 = pets.CrossEntropyDirectOptimizationSolver(mu0, bad, num_control_samples=S)
    assert L.rat_pets_sample_controls(C.byref(ds2.c), nv.P(nv.f64(zc)), nv.P(ctrl)) == 1
