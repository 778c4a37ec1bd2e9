"""The NumPy model of the block-form backward step (tests/wide32_model.py: the algorithm csrc/wide32.h implements for n, m <= 32 -- padded
blocks of 16 x 16 tiles, -M^-1 AND -H^-1 by the symmetric sweep with 2 x 2 block pivots run round by round as the kernel runs it, rounds
beyond the matrix skipped, positive definiteness from the blocks' leading minors, logdet from their determinants) reproduces the oracle's
gain sweep and policy evaluation: the design of the kernel is checked on the CPU, without a GPU."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc
from wide32_model import sweep, sweep_inverse


def test_sweep_inverse_is_minus_the_inverse_with_padding_and_odd_sizes():
    rng = np.random.default_rng(0)
    for n, p in ((32, 32), (20, 32), (17, 32), (5, 16), (1, 16), (16, 16)):
        G = rng.standard_normal((n, n))
        M = np.zeros((p, p)); M[:n, :n] = 3.0 * np.eye(n) + G @ G.T / n
        M[np.arange(n, p), np.arange(n, p)] = 1.0
        Mi, pd, ld = sweep_inverse(M, n)
        assert pd and np.abs(Mi[:n, :n] + np.linalg.inv(M[:n, :n])).max() < 1e-12 and abs(ld - np.linalg.slogdet(M[:n, :n])[1]) < 1e-10
        M[0, 0] = -1.0
        assert not sweep_inverse(M, n)[1]                       # a negative leading minor is seen


@pytest.mark.parametrize("n,m,N,kappa", [(20, 6, 10, 0.02), (24, 24, 6, 0.0), (32, 32, 4, 0.0), (17, 17, 8, 0.01), (5, 20, 8, 0.0), (32, 1, 8, 0.02)])
def test_wide32_model_matches_the_oracle(n, m, N, kappa):
    prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=5, kappa=kappa)
    P = orc.Problem(prob)
    u = 0.1 * np.random.default_rng(2).standard_normal((N, m))
    _, x = orc.simulate_open(P, x0, u)
    _, ap = orc.approximate_model(P, u, x)
    a = ap.arrays()
    Wk = prob.W
    for theta in (0.0, 0.5, 2.0):
        rc, Lo, dlo, dpo, _, _ = orc.dp_gain(P, ap, theta)
        Lm, dlm, s0, S0, why = sweep(a, N, Wk, theta, 0.0)
        assert rc == 0 and why is None
        e = [np.abs(Lm - Lo).max() / np.abs(Lo).max(), np.abs(dlm - dlo).max() / np.abs(dlo).max(),
             abs(s0 - dpo["s"][0]) / abs(dpo["s"][0]), np.abs(S0 - dpo["S"][0]).max() / np.abs(dpo["S"][0]).max()]
        _, dpe = orc.dp_eval(P, ap, Lo * 0.9, None, theta, 1e-6)
        _, _, se, _, _ = sweep(a, N, Wk, theta, 1e-6, L=Lo * 0.9)
        e.append(abs(se - dpe["s"][0]) / abs(dpe["s"][0]))
        assert max(e) < 1e-10, (theta, e)
    rc, *_ = orc.dp_gain(P, ap, 1e6)
    assert rc == 2 and sweep(a, N, Wk, 1e6, 0.0)[4] == "M"          # isposdef(M) fails in both
