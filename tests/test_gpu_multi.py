"""rat_multi -- several devices behind the C ABI, one calling thread -- on this one-GPU box: n_devices = 1 with and without the RCCL
collective (one-rank communicator, test hook), and n_devices = 2, 3, 8 LOGICAL devices (RATILQR_MULTI_LOGICAL=1: several handles and
streams on the one GPU, the all-gather carried out by stream-ordered device copies into the slots RCCL would fill) -- every line of the
G > 1 code but the ncclAllGather call itself -- against the single-handle entry points (bit for bit) and the oracle."""
import ctypes as C

import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from ratilqr.jl_amd import _native as nv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("force_rccl", [False, True])
def test_multi_compute_cost_and_ce_solve_equal_the_single_handle_path(force_rccl, monkeypatch):
    if force_rccl:
        monkeypatch.setenv("RATILQR_MULTI_FORCE_RCCL", "1")
    prob, x0, u = rat.synthetic_lq_problem()
    B = 96
    mc = rat.MultiContext(prob, max_batch=B, devices=(0,))
    assert mc.n_devices == 1 and mc.uses_rccl == force_rccl
    theta = np.concatenate([np.abs(1.0 + 2.0 * np.random.default_rng(3).standard_normal(B - 2)), [30.0, 50.0]])
    cost = mc.compute_cost(x0, u, theta, 0.1)
    ctx = rat.Context(prob, max_batch=B)
    v, st, _, _ = ctx.solve_batch(x0, u, theta)
    assert np.array_equal(cost, v + 0.1 / theta) and np.isposinf(cost[-1]) and st[-1] == 1
    assert mc.allgathers == (1 if force_rccl else 0)
    cost_small = mc.compute_cost(None, None, theta[:7], 0.1) if False else mc.compute_cost(x0, u, theta[:7], 0.1)     # smaller batch, same object
    assert np.array_equal(cost_small, cost[:7])
    # the whole RAT iLQR solve through rat_multi_ce_solve == rat_ce_solve on one handle, same injected stream
    z = np.random.default_rng(11).standard_normal(20000)
    ref_solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=B, num_elite=12)
    ref = ce.solve_(ref_solver, prob, x0, u, z, kl_bound=0.1)
    c = nv.CeSolver()
    nv.lib().rat_ce_default(C.byref(c))
    c.num_samples, c.num_elite = B, 12
    mc.set_stream(z)
    got = mc.ce_solve(c, x0, u, 0.1)
    assert got[0] == ref[0] and got[4] == ref[4] and got[5] == ref[5] and got[6] == ref[6] and np.array_equal(got[3], ref[3])
    assert c.mu_init == ref_solver.c.mu_init and c.n_solves == ref_solver.c.n_solves == 5 * B
    assert mc.allgathers == ((2 + 5) if force_rccl else 0)


def test_multi_rejects_more_devices_than_visible():
    import torch
    prob, _, _ = rat.synthetic_lq_problem()
    with pytest.raises(rat.RatError):
        rat.MultiContext(prob, max_batch=8, devices=tuple(range(torch.cuda.device_count() + 1)))
    with pytest.raises(rat.RatError):
        rat.MultiContext(prob, max_batch=8, devices=(0, 0))


def test_multi_pets_cost_equals_the_single_handle_cost():
    """rat_multi_pets_compute_cost (control samples sharded over devices, here one) == rat_pets_compute_cost, with injected noise and with
    the device generator -- which is keyed by the GLOBAL trajectory index: a block evaluated alone with sample0 > 0 through the Python
    sharding layer draws what it draws as part of the whole batch."""
    from ratilqr.jl_amd import pets, multi
    rng = np.random.default_rng(5)
    Nh, n, m, S, K = 30, 12, 4, 40, 25
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    prob = rat.LQGenerativeProblem(0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), Nh, ("gaussian", np.zeros(n), 0.05 * np.eye(n)),
                                   Q=np.eye(n), R=0.1 * np.eye(m), Qf=np.eye(n))
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, m)), np.stack([np.eye(m)] * Nh), num_control_samples=S, num_trajectory_samples=K)
    ctrl, x0 = 0.3 * rng.standard_normal((S, Nh, m)), rng.standard_normal(n)
    zn, zu = pets.draw_noise(prob, rng, S, K)
    mp = multi.MultiPetsContext(prob, devices=(0,))
    ref = pets.compute_cost_serial(ds, prob, x0, ctrl, None, streams=(zn, zu))
    assert np.array_equal(mp.compute_cost(x0, ctrl, K, streams=(zn, zu)), ref)
    dev = mp.compute_cost(x0, ctrl, K, seed=7)
    assert np.array_equal(dev, pets.compute_cost_serial(ds, prob, x0, ctrl, None, seed=7)) and np.all(np.isfinite(dev))
    assert np.array_equal(dev, mp.compute_cost(x0, ctrl, K, seed=7)) and not np.array_equal(dev, mp.compute_cost(x0, ctrl, K, seed=8))


def test_multi_shards_on_the_round_based_path_run_on_helper_threads(monkeypatch):
    """E = 8 with a shard beyond one generation of workgroups (300 > 256 samples) runs the round-based path, whose host loop polls the
    device: rat_multi hands such shards to a helper thread per device (joined before the collective) -- same costs as the single handle"""
    monkeypatch.setenv("RATILQR_MULTI_FORCE_RCCL", "1")
    prob, x0, u = rat.synthetic_lq_problem()
    B = 300
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(8).standard_normal(B))
    theta[-1] = 80.0
    mc = rat.MultiContext(prob, max_batch=B, spec_eps=8, devices=(0,))
    cost = mc.compute_cost(x0, u, theta, 0.1)
    ctx = rat.Context(prob, max_batch=B, spec_eps=8)
    ctx.profile(True)
    v, st, _, _ = ctx.solve_batch(x0, u, theta)
    assert "solve_block" not in [k for k, p in ctx.profile_get().items() if p["launches"]]      # (the round-based path indeed)
    assert np.array_equal(cost[:-1], (v + 0.1 / theta)[:-1]) and np.isposinf(cost[-1]) and st[-1] == 1 and mc.allgathers == 1
    assert np.array_equal(mc.compute_cost(x0, u, theta[:100], 0.1), cost[:100])                    # a block-kernel shard right after


@pytest.mark.parametrize("G", [2, 3, 8])
def test_logical_devices_run_the_multi_device_code(G, monkeypatch):
    """n_devices > 1 on a one-GPU box (logical devices): contiguous ragged shards (B = 10, 13 on 8 devices: empty and short blocks whose
    pad slots must never be read), one gather per batch, and the gathered status / iteration / line-search counts of every shard equal
    the single handle's, as do the costs, bit for bit; then the whole CE solve."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    monkeypatch.setenv("RATILQR_MULTI_LOGICAL", "1")
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.03)
    rng = np.random.default_rng(20 + G)
    Bmax = 1024
    mc = rat.MultiContext(prob, max_batch=Bmax, devices=tuple(range(G)))
    assert mc.n_devices == G and mc.is_logical and not mc.uses_rccl
    ctx = rat.Context(prob, max_batch=Bmax)
    n_g = 0
    for B in (10, 13, 1024, 5, 1):
        theta = np.abs(1.0 + 2.0 * rng.standard_normal(B))
        theta[B // 2] = 70.0                                                # an infeasible sample in some shard
        if B > 3:
            theta[1] = 0.0                                                  # theta = 0: cost = value + kl / 0 = Inf, status OK
        v, st, it, ls = ctx.solve_batch(x0, u, theta)
        cost, st_m, it_m, ls_m = mc.compute_cost_ex(x0, u, theta, 0.1)
        n_g += 1
        with np.errstate(divide="ignore"):
            assert np.array_equal(cost, v + 0.1 / theta)
        assert np.array_equal(st_m, st) and np.array_equal(it_m, it) and np.array_equal(ls_m, ls)
        assert st_m[B // 2] == 1 and np.isposinf(cost[B // 2]) and (st_m >= 0).all()          # (pad slots hold -1 / NaN: never seen)
        assert mc.allgathers == n_g
        v_m, st_v, it_v, ls_v = mc.solve_batch(x0, u, theta)
        n_g += 1
        assert np.array_equal(v_m, v) and np.array_equal(st_v, st) and np.array_equal(it_v, it) and np.array_equal(ls_v, ls)
    # the whole RAT iLQR solve through rat_multi_ce_solve == rat_ce_solve on one handle, same injected stream
    z = np.random.default_rng(11).standard_normal(20000)
    ref_solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=100, num_elite=12)
    ref = ce.solve_(ref_solver, prob, x0, u, z, kl_bound=0.1)
    c = nv.CeSolver()
    nv.lib().rat_ce_default(C.byref(c))
    c.num_samples, c.num_elite = 100, 12
    mc.set_stream(z)
    got = mc.ce_solve(c, x0, u, 0.1)
    assert got[0] == ref[0] and got[4] == ref[4] and got[5] == ref[5] and got[6] == ref[6] and np.array_equal(got[3], ref[3])
    assert c.n_solves == ref_solver.c.n_solves and mc.allgathers == n_g + c.n_solves // 100


def test_logical_devices_with_polled_shards(monkeypatch):
    """Speculation width 8 on 2 logical devices with shards of 300 samples: every shard runs the round-based path on its own helper thread
    (two host loops polling two streams of the one GPU at once); statuses and counts come back with the costs."""
    monkeypatch.setenv("RATILQR_MULTI_LOGICAL", "1")
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.05)
    B = 600
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(9).standard_normal(B))
    theta[17] = 90.0
    mc = rat.MultiContext(prob, max_batch=B, spec_eps=8, devices=(0, 1))
    cost, st_m, it_m, ls_m = mc.compute_cost_ex(x0, u, theta, 0.1)
    ctx = rat.Context(prob, max_batch=B, spec_eps=8)
    assert ctx.get_path(300) == "rounds" and ctx.get_path(B) == "rounds"
    v, st, it, ls = ctx.solve_batch(x0, u, theta)
    assert np.array_equal(cost, v + 0.1 / theta) and np.array_equal(st_m, st) and np.array_equal(it_m, it) and np.array_equal(ls_m, ls)
    assert st[17] == 1 and it.max() >= 3 and mc.allgathers == 1


def test_set_path_per_handle(monkeypatch):
    """rat_set_path: the execution path is a property of the handle, not of the process environment; results do not depend on it."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.04)
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(4).standard_normal(64)); theta[5] = 60.0
    ctx = rat.Context(prob, max_batch=64)
    assert ctx.get_path(64) == "block"
    ref = ctx.solve_batch(x0, u, theta)
    for path in ("fused", "rounds", "block", "auto"):
        ctx.set_path(path)
        assert ctx.get_path(64) == (path if path != "auto" else "block")
        ctx.profile(True)
        got = ctx.solve_batch(x0, u, theta)
        kinds = {k for k, p in ctx.profile_get().items() if p["launches"]}
        ctx.profile(False); ctx.profile_reset()
        assert ("solve_fused" in kinds) == (path == "fused") and ("solve_block" in kinds) == (path in ("block", "auto")), (path, kinds)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), path
    c8 = rat.Context(prob, max_batch=64, spec_eps=8)
    assert c8.get_path(64) == "block"
    with pytest.raises(rat.RatError):
        c8.set_path("fused")                                                # no one-wavefront kernel for E = 8
    c8.set_path("rounds")
    got = c8.solve_batch(x0, u, theta)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    with pytest.raises(rat.RatError):
        rat.Context(prob, max_batch=4, spec_eps=3).set_path("block")


def test_default_switch_shards_of_the_baseline_batch_against_the_oracle(monkeypatch):
    """What an 8-GPU run of the BASELINE batch runs: B = 1024 over G = 8 logical devices with DEFAULT switches, so
    every shard is a 128-sample launch of the time-parallel workgroup-per-sample kernel (block_psw on) -- through
    rat_multi_ce_compute_cost_ex against the ORACLE (compute_cost, cross_entropy_bilevel_optimization.jl:173-195: identical status /
    iteration / line-search counts, costs to 1e-9), and the whole rat_multi_ce_solve against the one-device run and the oracle's CE
    solve on the same injected stream (theta_opt, mu, sigma to 1e-9; SURVEY App. B.16: a result must not depend on the sharding)."""
    import os
    from oracle import oracle as orc
    monkeypatch.delenv("RATILQR_BLOCK_PSW", raising=False)
    monkeypatch.setenv("RATILQR_MULTI_LOGICAL", "1")
    G, B = 8, 1024
    prob, x0, u = rat.synthetic_lq_problem()
    mc = rat.MultiContext(prob, max_batch=B, devices=tuple(range(G)))
    assert mc.n_devices == G and mc.is_logical
    for i in range(G):                                                      # every shard: the time-parallel kernel, by the handle's own word
        v = C.c_int64()
        nv.check(nv.lib().rat_debug_get(mc.handle(i), b"block_psw", C.byref(v)))
        assert v.value == 1
    rng = np.random.default_rng(77)
    theta = np.abs(1.0 + 2.0 * rng.standard_normal(B))
    theta[5], theta[300], theta[1023] = 0.0, 70.0, 13.4                    # theta = 0, infeasible in initialize!, near the breakdown
    cost, st, it, ls = mc.compute_cost_ex(x0, u, theta, 0.1)
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=os.cpu_count() or 8)
    assert np.array_equal(st, so) and np.array_equal(it, io) and np.array_equal(ls, lo)
    with np.errstate(divide="ignore"):
        co = vo + 0.1 / theta
    fin = np.isfinite(co)
    assert np.array_equal(fin, np.isfinite(cost)) and fin.sum() > 1000
    assert (np.abs(cost[fin] - co[fin]) <= 1e-9 * np.abs(co[fin])).all()
    # the same batch on ONE device (1024 samples: the one-wavefront kernel, another rounding order): to rounding
    ctx = rat.Context(prob, max_batch=B)
    v1, s1, i1, l1 = ctx.solve_batch(x0, u, theta)
    assert np.array_equal(s1, st) and np.array_equal(i1, it) and np.array_equal(l1, ls)
    with np.errstate(divide="ignore"):
        assert (np.abs(cost[fin] - (v1 + 0.1 / theta)[fin]) <= 1e-12 * np.abs(co[fin])).all()
    # the whole RAT iLQR solve: 8 shards vs one device vs the oracle, same injected stream
    z = np.random.default_rng(31).standard_normal(40000)
    kw = dict(num_samples=B, num_elite=100)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, nthreads=os.cpu_count() or 8, **kw)
    rc, th_o, x_o, l_o, L_o, val_o, tmin_o, tmax_o = oc.solve(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0
    one = rat.CrossEntropyBilevelOptimizationSolver(**kw)
    th1, x1, l1_, L1, val1, tmin1, tmax1 = ce.solve_(one, prob, x0, u, z, kl_bound=0.1)
    c = nv.CeSolver()
    nv.lib().rat_ce_default(C.byref(c))
    c.num_samples, c.num_elite = B, 100
    mc.set_stream(z)
    n_g = mc.allgathers
    th, x, l, L, val, tmin, tmax = mc.ce_solve(c, x0, u, 0.1)
    assert mc.allgathers == n_g + c.n_solves // B and c.n_solves == one.c.n_solves == 5 * B
    for a, b in ((th, th_o), (val, val_o), (c.mu, oc.c.mu), (th, th1), (val, val1), (c.mu, one.c.mu)):
        assert abs(a - b) <= 1e-9 * abs(b)
    assert abs(c.sigma - oc.c.sigma) <= 1e-6 * oc.c.sigma + 1e-12 and abs(c.sigma - one.c.sigma) <= 1e-6 * one.c.sigma + 1e-12
    assert tmin == tmin_o == tmin1 and tmax == tmax_o == tmax1             # extreme samples: identical draws
    assert c.mu_init == oc.c.mu_init == one.c.mu_init and c.sigma_init == oc.c.sigma_init
    assert np.abs(x - x_o).max() < 1e-9 and np.abs(L - L_o).max() < 1e-9 and np.abs(l - l_o).max() < 1e-9
