"""rat_multi -- several devices behind the C ABI, one calling thread -- on this one-GPU box: n_devices = 1 with and without the RCCL
collective (one-rank communicator, test hook), against the single-handle entry points (bit for bit) and the oracle.  The shard
arithmetic for n_devices > 1 is covered on CPU (tests/test_cpu_abi.py, tests/test_cpu_distributed.py)."""
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
