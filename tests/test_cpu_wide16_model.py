"""The NumPy model of the register-form backward step (tests/wide16_model.py: the algebra csrc/wide16.h implements for n <= 16, m <= 4 --
two-tile products, X = S [A | B | S^-1 s_vec], Y = theta M^-1 X, T = X + S Y, G read out of A'T2's columns, padded unit pivots) reproduces
the oracle's gain sweep and policy evaluation: the design of the kernel is checked on the CPU, without a GPU."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc
from wide16_model import sweep


@pytest.mark.parametrize("n,m,N,kappa", [(16, 4, 12, 0.02), (13, 1, 10, 0.0), (15, 3, 8, 0.0), (14, 2, 10, 0.03)])
def test_wide16_model_matches_the_oracle(n, m, N, kappa):
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
