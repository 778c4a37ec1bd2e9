"""The NumPy model of the TIME-PARALLEL Riccati sweep (tests/psweep_model.py: what csrc/psweep.h implements -- segment elements in the
conditional-value-function form of the associative LQ scan, information-form hops, the ordinary step from the true boundary values)
reproduces the sequential step model (tests/step_model.py, itself held against the oracle by tests/test_cpu_step_model.py) and the oracle:
gains, value and the value function at time 0 to rounding, for gain sweeps and policy evaluations, 2 ... 8 waves, padded sizes."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc
from step_model import sweep, AUG
from psweep_model import psweep, boundaries


def approx(n, m, N, kappa):
    prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=3, kappa=kappa)
    P = orc.Problem(prob)
    u = 0.1 * np.random.default_rng(1).standard_normal((N, m))
    _, x = orc.simulate_open(P, x0, u)
    _, ap = orc.approximate_model(P, u, x)
    return prob, P, ap, ap.arrays()


def test_cut_model():
    for N, P in ((50, 2), (50, 4), (50, 8), (20, 4), (10, 4), (5, 4)):
        c = boundaries(N, P)
        assert c[0] == 0 and c[-1] == N and all(b > a for a, b in zip(c[:-2], c[1:-1])) and len(c) <= P + 2
    assert boundaries(50, 4) == [0, 7, 16, 25, 36, 50] and boundaries(50, 2) == [0, 14, 30, 50]


@pytest.mark.parametrize("n,m,N,kappa", [(12, 4, 50, 0.0), (12, 4, 50, 0.05), (4, 2, 20, 0.0), (7, 3, 33, 0.02)])
def test_segment_parallel_sweep_equals_the_sequential_sweep(n, m, N, kappa):
    prob, P, ap, a = approx(n, m, N, kappa)
    for theta in (0.5, 6.0, 11.0):
        Ls, dls, s, V = sweep(a, n, m, N, prob.W(0), theta, 0.0)
        _, Lo, dlo, dpo, _, _ = orc.dp_gain(P, ap, theta)
        Le = 0.9 * Ls
        _, _, se, _ = sweep(a, n, m, N, prob.W(0), theta, 1e-6, L=Le)
        for waves in (2, 3, 4, 8):
            cuts = boundaries(N, waves)
            Lp, dlp, V0, _, ok = psweep(a, n, m, N, prob.W(0), theta, 0.0, cuts)
            assert ok
            assert np.abs(Lp - Ls).max() <= 1e-12 * np.abs(Ls).max() and np.abs(dlp - dls).max() <= 1e-12 * np.abs(dls).max()
            assert np.abs(Lp - Lo).max() <= 1e-10 * np.abs(Lo).max()                                     # ... and the oracle's gains
            assert abs(V0[AUG, AUG] / 2 - s[0]) <= 1e-12 * abs(s[0]) and abs(V0[AUG, AUG] / 2 - dpo["s"][0]) <= 1e-10 * abs(s[0])
            assert np.abs(V0[:n, :n] - V[:n, :n]).max() <= 1e-12 * np.abs(V[:n, :n]).max()
            for noise_form in (False, True):
                _, _, V0e, _, oke = psweep(a, n, m, N, prob.W(0), theta, 1e-6, cuts, L=Le, noise_form=noise_form)
                assert oke and abs(V0e[AUG, AUG] / 2 - se[0]) <= 1e-12 * abs(se[0])


def test_values_handed_along_the_chain_are_the_sequential_ones():
    """the hop's output at a cut against the sequential recursion's value there (S and s_vec; the additive scalar is not propagated)"""
    from step_model import step, pad_tiles, NP
    from psweep_model import terminal
    n, m, N = 12, 4, 50
    prob, P, ap, a = approx(n, m, N, 0.02)
    theta = 9.0
    cuts = boundaries(N, 4)
    Winv = np.linalg.inv(prob.W(0)); Wp = prob.W(0)
    V = terminal(a, n, N); seq = {}
    for t in reversed(range(N)):
        V, _, ok1, ok2 = step(V, *pad_tiles(a, t, n, m), Winv, Wp, np.linalg.slogdet(prob.W(0))[1], theta, 0.0, n, m)
        assert ok1 and ok2
        seq[t] = V.copy()
    _, _, _, Vb, ok = psweep(a, n, m, N, prob.W(0), theta, 0.0, cuts)
    assert ok and sorted(Vb) == [1, 2, 3, 4]
    for s, Vh in Vb.items():
        Vs = seq[cuts[s]]
        assert np.abs(Vh[:n, :n] - Vs[:n, :n]).max() <= 1e-13 * np.abs(Vs[:n, :n]).max()
        assert np.abs(Vh[:n, AUG] - Vs[:n, AUG]).max() <= 1e-13 * np.abs(Vs[:n, AUG]).max()
