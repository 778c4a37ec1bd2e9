"""The time-parallel closed-loop rollout (csrc/kernels.hip: rollprl_body) as a NumPy model (tests/rollprl_model.py) against the sequential
recursion of simulate_dynamics (ileqg.jl:62-87), and the properties of the host's cut model -- no GPU needed."""
import numpy as np
import pytest

import rollprl_model as rm


def _problem(n, m, N, seed):
    rng = np.random.default_rng(seed)
    A = 0.9 * np.linalg.qr(rng.standard_normal((n, n)))[0]
    B = rng.standard_normal((n, m)) / np.sqrt(n)
    L = 0.3 * rng.standard_normal((N, m, n))
    l, dl = 0.2 * rng.standard_normal((N, m)), rng.standard_normal((N, m))
    x0 = rng.standard_normal(n)
    xbar = np.zeros((N + 1, n)); xbar[0] = x0                    # (xbar, l): a trajectory of the same dynamics from the same x_0
    for t in range(N):
        xbar[t + 1] = A @ xbar[t] + B @ l[t]
    return A, B, L, l, dl, xbar, x0


@pytest.mark.parametrize("n,m,N", [(12, 4, 50), (12, 4, 16), (7, 3, 33), (3, 1, 52), (12, 4, 600)])
def test_segments_and_hops_reproduce_the_sequential_rollout(n, m, N):
    A, B, L, l, dl, xbar, x0 = _problem(n, m, N, N)
    for eps in (1.0, 0.25, 1e-3):
        xs, us = rm.sequential(A, B, L, l, dl, xbar, x0, eps)
        for c in (rm.cuts(N), [0, 1, 2, 3, N], [0, N - 3, N - 2, N - 1, N], rm.cuts(N, 4.0, 0.0, 0.0)):
            xp, up = rm.time_parallel(A, B, L, l, dl, xbar, x0, eps, c)
            scale = max(1.0, np.abs(xs).max())
            assert np.abs(xp - xs).max() <= 1e-12 * scale and np.abs(up - us).max() <= 1e-12 * scale, (c, eps)


def test_the_affine_part_scales_with_eps_and_the_linear_part_does_not():
    """What would let the maps be kept across line-search candidates of one gain sweep (built ahead, measured, dropped)."""
    A, B, L, l, dl, xbar, x0 = _problem(12, 4, 30, 1)
    P1, c1 = rm.segment_map(A, B, L, dl, 5, 19, 1.0)
    P2, c2 = rm.segment_map(A, B, L, dl, 5, 19, 0.125)
    assert np.array_equal(P1, P2) and np.allclose(c2, 0.125 * c1, rtol=1e-14, atol=0)


def test_cut_model_is_monotone_balanced_and_complete():
    for N in list(range(16, 70)) + [100, 257, 600]:
        for e, h, epi in ((0.45, 0.9, 1.0), (0.01, 0.0, 0.0), (4.0, 0.0, 0.0), (0.1, 20.0, 0.0), (0.45, 0.9, 15.0), (0.99, 3.0, 3.0)):
            c = rm.cuts(N, e, h, epi)
            assert c[0] == 0 and c[4] == N and all(c[w] < c[w + 1] for w in range(4)), (N, e, h, epi, c)
    for N in (50, 52, 100, 257):                                 # the default model: every wave done within ~a step and a half of the others
        f = rm.finish_times(rm.cuts(N))
        assert max(f) - min(f) <= 1.6, (N, f)
    assert rm.cuts(50) == [0, 21, 32, 42, 50]                    # (the cuts the device reports for the BASELINE horizon: profiles/r06_duo.md)
    # against the sequential chain: N ordinary steps -> the longest wave
    assert max(rm.finish_times(rm.cuts(50))) < 0.45 * 50
