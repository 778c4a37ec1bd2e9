"""State the host mirrors carry between calls: solvers follow their `problem` argument, a handle can be re-bound to
another problem, the CE `rng` argument behaves like the reference's stateful AbstractRNG, generic-closure contexts refuse the
device-family entry points."""
import ctypes as C

import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import cross_entropy as ce
from ratilqr.jl_amd import _native as nv

pytestmark = pytest.mark.gpu


def test_ileqg_solver_follows_the_problem_argument():
    """solve!(ileqg, problem, ...) takes every table from `problem` (ileqg.jl:635-659): a solver constructed on one problem and called
    with another solves the other one (round 1 silently solved the constructor's)."""
    pa, x0a, ua = rat.synthetic_lq_problem(seed=0)
    pb, x0b, ub = rat.synthetic_lq_problem(n=6, m=2, N=30, seed=3)
    s = rat.ILEQGSolver(pa)
    ra = rat.solve_(s, pa, x0a, ua, 1.5)
    rb = rat.solve_(s, pb, x0b, ub, 1.5)                     # same solver object, other problem
    fresh = rat.solve_(rat.ILEQGSolver(pb), pb, x0b, ub, 1.5)
    assert rb[3] == fresh[3] and np.array_equal(rb[0], fresh[0]) and np.array_equal(rb[2], fresh[2])
    assert rb[0].shape == (31, 6) and ra[0].shape == (51, 12)
    ra2 = rat.solve_(s, pa, x0a, ua, 1.5)                    # and back
    assert ra2[3] == ra[3] and np.array_equal(ra2[2], ra[2])
    # the stepwise operators follow it too
    s2 = rat.ILEQGSolver(pa)
    rat.initialize_ileqg_(s2, pb, x0b, ub, 0.7)
    s3 = rat.ILEQGSolver(pb)
    rat.initialize_ileqg_(s3, pb, x0b, ub, 0.7)
    assert s2.value_current == s3.value_current


def test_handle_rebound_to_a_smaller_problem_with_the_same_horizon():
    """rat_problem_set on a live handle: same N, smaller n and m -- the padded lanes of the slot pools must read as zeros again."""
    big, x0, u = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=0)
    small, x0s, us = rat.synthetic_lq_problem(n=5, m=2, N=50, seed=4, w=1e-2)
    theta = np.array([0.0, 0.4, 1.0, 2.0])
    for E in (1, 2):
        ctx = rat.Context(big, max_batch=8, spec_eps=E)
        vb = ctx.solve_batch(x0, u, theta)
        ctx.set_problem(small)
        got = ctx.solve_batch(x0s, us, theta)
        want = rat.Context(small, max_batch=8, spec_eps=E).solve_batch(x0s, us, theta)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        r, rw = ctx.solve(x0s, us, 0.4), rat.Context(small, spec_eps=E).solve(x0s, us, 0.4)
        assert r["value"] == rw["value"] and np.array_equal(r["L"], rw["L"]) and np.array_equal(r["x"], rw["x"])
        ctx.set_problem(big)                                   # and back to the large one
        vb2 = ctx.solve_batch(x0, u, theta)
        for a, b in zip(vb, vb2):
            assert np.array_equal(a, b)


def test_ce_rng_argument_is_stateful():
    prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
    # (a) an integer seed names ONE generator: a hand-driven step_ loop advances it instead of replaying it
    s = rat.CrossEntropyBilevelOptimizationSolver(num_samples=8, num_elite=3)
    ce.initialize_(s)
    th1, _ = ce.step_(s, prob, x0, u, 0.1, 42)
    th2, _ = ce.step_(s, prob, x0, u, 0.1, 42)
    assert not np.array_equal((th1 - s.c.mu_init), (th2 - s.c.mu_init)) and s.c.iter_current == 2
    s_b = rat.CrossEntropyBilevelOptimizationSolver(num_samples=8, num_elite=3)
    ce.initialize_(s_b)
    th1b, _ = ce.step_(s_b, prob, x0, u, 0.1, 42)
    assert np.array_equal(th1, th1b)                           # same seed, same first batch
    # (b) a Python list is converted once and keyed on the caller's object: the second step_ continues in it
    z = np.random.default_rng(3).standard_normal(4000)
    zl = z.tolist()
    sl = rat.CrossEntropyBilevelOptimizationSolver(num_samples=8, num_elite=3)
    ce.initialize_(sl)
    a1, _ = ce.step_(sl, prob, x0, u, 0.1, zl)
    a2, _ = ce.step_(sl, prob, x0, u, 0.1, zl)
    sa = rat.CrossEntropyBilevelOptimizationSolver(num_samples=8, num_elite=3)
    ce.initialize_(sa)
    b1, _ = ce.step_(sa, prob, x0, u, 0.1, z)
    b2, _ = ce.step_(sa, prob, x0, u, 0.1, z)
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and not np.array_equal(a1, a2)
    # (c) the Context is recreated between two steps (a compute_cost batch larger than max_batch): the stream resumes, it is neither
    #     lost (RAT_ERR_STREAM_DRY in round 1) nor restarted
    sc = rat.CrossEntropyBilevelOptimizationSolver(num_samples=8, num_elite=3)
    ce.initialize_(sc)
    c1, _ = ce.step_(sc, prob, x0, u, 0.1, z)
    old = sc.context(prob)
    ce.compute_cost(sc, prob, x0, u, np.linspace(0.1, 1.0, 20), 0.1)      # 20 > 8: new Context
    assert sc.context(prob) is not old
    c2, _ = ce.step_(sc, prob, x0, u, 0.1, z)
    assert np.array_equal(c1, b1) and np.array_equal(c2, b2)
    assert sc._stream_off + nv.lib().rat_ce_stream_pos(sc.context(prob).h) == nv.lib().rat_ce_stream_pos(sa.context(prob).h)


def test_generic_context_refuses_device_family_entry_points():
    n, m, N = 2, 1, 8
    A, B = np.array([[1.0, 0.1], [0.0, 1.0]]), np.array([[0.0], [0.1]])
    gp = rat.GenericRiskSensitiveProblem(lambda x, u: A @ x + B @ u, lambda k, x, u: 0.5 * x @ x + 0.5 * u @ u, lambda x: 0.5 * x @ x,
                                         lambda k: 0.01 * np.eye(n), N, n, m)
    ctx = rat.ileqg.make_context(gp)
    for call in (lambda: ctx.rollout_noisy(np.zeros(n), np.zeros((N, m)), K=2), lambda: ctx.solve_batch(np.zeros(n), np.zeros((N, m)), [0.1]),
                 lambda: ctx.solve_batch_dev(0, 1, 0), lambda: ctx.compute_cost_dev(0, 1, 0.1, 0), lambda: ctx.set_initial(np.zeros(n), np.zeros((N, m)))):
        with pytest.raises(NotImplementedError):
            call()
    with pytest.raises(NotImplementedError):
        rat.simulate_dynamics_noisy(gp, np.zeros(n), np.zeros((N, m)), K=2)


def test_eps_history_is_never_truncated():
    """the reference's eps_history is unbounded (ileqg.jl:537): a buffer that is too small is grown and the solve repeated"""
    import numpy as np
    import ratilqr.jl_amd as rat
    prob, x0, u = rat.synthetic_lq_problem(seed=5, kappa=0.05)
    ctx = rat.Context(prob)
    full = ctx.solve(x0, u, 5.0)
    assert full["hist_n"] >= 4
    small = ctx.solve(x0, u, 5.0, hist_cap=2)
    assert small["hist_n"] == full["hist_n"] and np.array_equal(small["eps_history"], full["eps_history"])


def test_handles_release_their_device_memory():
    """create / bind / solve / re-bind / destroy in a loop (tile-sized, general-size, speculative, multi-device, PETS handles): the free
    device memory afterwards is what it was before"""
    import gc
    import torch
    import ratilqr.jl_amd as rat

    def cycle(k):
        prob, x0, u = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=k)
        big, bx0, bu = rat.synthetic_lq_problem(n=20, m=6, N=20, seed=k)
        th = np.array([0.0, 0.5, 1.0])
        ctx = rat.Context(prob, max_batch=64, spec_eps=(1, 2, 8)[k % 3])
        ctx.solve_batch(x0, u, th)
        ctx.solve(x0, u, 0.5)
        ctx.set_problem(big)                              # tile-sized -> general-size pools on the same handle
        ctx.solve_batch(bx0, bu, 0.2 * th)
        ctx.rollout_open(bx0, bu)
        ctx.set_problem(prob)
        ctx.solve_batch(x0, u, th)
        mc = rat.MultiContext(prob, max_batch=16, devices=(0,))
        mc.compute_cost(x0, u, th + 0.1, 0.1)
        del ctx, mc

    for k in range(3):                                    # warm-up: allocator pools, code objects, RCCL (if loaded) settle
        cycle(k)
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(30):
        cycle(k)
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 30 create / destroy cycles"


@pytest.mark.parametrize("B", [96, 640])
def test_shared_initial_trajectory_follows_x0_and_u(B, monkeypatch):
    """initialize!'s open-loop rollout (ileqg.jl:225-228) is rolled out once per (x_0, u_array) and copied by the samples of every batch
    (FusedArgs.init_*): a handle that is given another x_0, another u_array or another problem must not reuse the old trajectory, and the
    results equal those of per-sample rollouts (RATILQR_INIT_SHARE=0) bit for bit.  B = 96: workgroup-per-sample kernel, 640: fused kernel."""
    prob, x0, u = rat.synthetic_lq_problem(seed=0)
    rng = np.random.default_rng(3)
    theta = np.abs(1.0 + 2.0 * rng.standard_normal(B)) + 1e-3
    x1, u1 = x0 * 0.7 + 0.1, u + 0.05 * rng.standard_normal(u.shape)
    prob2, _, _ = rat.synthetic_lq_problem(seed=5)
    ctx = rat.Context(prob, max_batch=B)
    seq = [(prob, x0, u), (prob, x1, u), (prob, x1, u1), (prob, x0, u), (prob2, x0, u), (prob, x0, u)]
    got = []
    for p, xa, ua in seq:
        if p is not ctx.problem:
            ctx.set_problem(p)
        got.append(ctx.solve_batch(xa, ua, theta))
    monkeypatch.setenv("RATILQR_INIT_SHARE", "0")
    for (p, xa, ua), g in zip(seq, got):
        ref = rat.Context(p, max_batch=B).solve_batch(xa, ua, theta)
        for a, b in zip(g, ref):
            assert np.array_equal(a, b)
    assert np.array_equal(got[0][0], got[3][0]) and np.array_equal(got[0][0], got[5][0])
    assert not np.array_equal(got[0][0], got[1][0]) and not np.array_equal(got[1][0], got[2][0]) and not np.array_equal(got[0][0], got[4][0])
