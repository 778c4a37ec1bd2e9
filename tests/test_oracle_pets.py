"""Pins the oracle's PETS restatement with the reference's known-answer test (/root/reference/test/pets_test.jl, lines
cited): manual recomputation of the rollout costs, elite selection, the smoothed mean / diagonal-variance update.  CPU only."""
import numpy as np

import ratilqr.jl_amd as rat
from oracle import oracle as orc

N = 20


def ref_problem():                                           # pets_test.jl:15-20: f = x + u + rand(rng, 2); c = sum(abs.(u)); h = 1
    return rat.LQGenerativeProblem(np.eye(2), np.eye(2), N, ("uniform", 0.0, 1.0), l1u=1.0, q0f=1.0)


def test_compute_cost_matches_manual_recomputation():         # :45-63
    prob = ref_problem()
    G = orc.GenProblem(prob)
    rng = np.random.default_rng(1234)
    S, K = 20, 100
    ctrl = rng.random((S, N, 2))
    zn = rng.random(S * K * N * 2)
    cost = orc.pets_compute_cost(G, np.zeros(2), ctrl, K, False, zn)
    # c does not depend on x (the reference test's remark at :50), so every rollout of a sample has the same cost
    for ii in range(S):
        x, c = np.zeros(2), 0.0
        for t in range(N):
            c += prob.c(t, x, ctrl[ii, t])
            x = x + ctrl[ii, t] + zn[((ii * K) * N + t) * 2: ((ii * K) * N + t) * 2 + 2]
        c += prob.h(x)
        assert np.isclose(c, cost[ii], rtol=1e-12)
    assert np.allclose(cost, np.abs(ctrl).sum(axis=(1, 2)) + 1.0, rtol=1e-12)


def test_state_dependent_cost_uses_every_rollout():
    rngp = np.random.default_rng(3)
    A = 0.9 * np.linalg.qr(rngp.standard_normal((4, 4)))[0]
    prob = rat.LQGenerativeProblem(A, rngp.standard_normal((4, 2)) / 2, 12, ("gaussian", 0.1 * np.ones(4), 0.05 * np.eye(4) + 0.01),
                                   Q=np.eye(4), R=0.1 * np.eye(2), Qf=2 * np.eye(4), kappa=-0.01, l1u=0.3,
                                   true_noise=(0.4, np.ones(4), 0.2 * np.eye(4)))
    G = orc.GenProblem(prob)
    S, K, n = 3, 5, 4
    ctrl = 0.3 * rngp.standard_normal((S, 12, 2))
    zn, zu = rngp.standard_normal(S * K * 12 * n), rngp.random(S * K * 12)
    x0 = rngp.standard_normal(4)
    for use_true in (False, True):
        cost = orc.pets_compute_cost(G, x0, ctrl, K, use_true, zn, zu)
        ref = np.zeros(S)
        for ii in range(S):
            for kk in range(K):
                j = ii * K + kk
                x, c = x0.copy(), 0.0
                for t in range(12):
                    c += prob.c(t, x, ctrl[ii, t])
                    z = zn[(j * 12 + t) * n:(j * 12 + t) * n + n]
                    xn = prob.lq.f(x, ctrl[ii, t])
                    if use_true and zu[j * 12 + t] < prob.tw2:
                        x = xn + prob.tmean2 + prob.tchol2 @ z
                    else:
                        x = xn + prob.nmean + prob.nchol @ z
                c += prob.h(x)
                ref[ii] += c / K
        assert np.all(np.isfinite(ref)) and np.allclose(cost, ref, rtol=1e-12)


def test_elites_and_distribution_update():                    # :66-84
    rng = np.random.default_rng(5)
    S, E = 20, 5
    s = orc.PetsSolver(np.zeros((N, 2)), np.stack([np.eye(2)] * N), num_control_samples=S, num_elite=E, smoothing_factor=0.1)
    ctrl = rng.random((S, N, 2))
    cost = rng.random(S)
    cost[3] = cost[7]                                          # a tie: stable sort keeps index order
    mu_old, Sig_old = s.mu_array, s.Sigma_array
    idx = s.update(ctrl, cost)
    assert np.array_equal(idx, np.argsort(cost, kind="stable")[:E])
    el = ctrl[idx]
    assert np.allclose(s.mu_array, 0.9 * el.mean(axis=0) + 0.1 * mu_old, rtol=1e-13)
    for t in range(N):
        assert np.allclose(s.Sigma_array[t], 0.9 * np.diag(el[:, t].var(axis=0, ddof=1)) + 0.1 * Sig_old[t], rtol=1e-12)


def test_step_and_initialize_counters():                      # :25-41, :87-94
    prob = ref_problem()
    G = orc.GenProblem(prob)
    rng = np.random.default_rng(1234)
    S, K = 20, 10
    s = orc.PetsSolver(np.zeros((N, 2)), np.stack([np.eye(2)] * N), num_control_samples=S, num_trajectory_samples=K, num_elite=5, iter_max=3)
    rc, ctrl, cost = s.step(G, np.zeros(2), False, rng.standard_normal(S * N * 2), rng.random(S * K * N * 2))
    assert rc == 0 and s.c.iter_current == 1
    assert np.allclose(ctrl.reshape(S, N, 2), ctrl) and not np.allclose(s.mu_array, 0)
    s.initialize()
    assert s.c.iter_current == 0 and np.all(s.mu_array == 0) and np.all(s.Sigma_array == np.eye(2))
