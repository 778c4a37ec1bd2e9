"""The NumPy model of the general-size backward step (tests/wide_step_model.py: the algebra csrc/wide.hip implements -- Cholesky of M,
one forward substitution, D S = S + theta Z'Z) reproduces the oracle's gain sweep and policy evaluation, and the first-principles
Gaussian integral of tests/leqg_exact.py: the design of the kernel is checked on the CPU, without a GPU."""
import numpy as np
import pytest

import leqg_exact as ex
import ratilqr.jl_amd as rat
from oracle import oracle as orc
from wide_step_model import sweep


@pytest.mark.parametrize("n,m,N,kappa", [(16, 5, 20, 0.02), (32, 8, 8, 0.0), (9, 9, 12, 0.0), (4, 2, 15, 0.0)])
def test_wide_step_model_matches_the_oracle(n, m, N, kappa):
    prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=3, kappa=kappa)
    P = orc.Problem(prob)
    u = 0.1 * np.random.default_rng(1).standard_normal((N, m))
    _, x = orc.simulate_open(P, x0, u)
    _, ap = orc.approximate_model(P, u, x)
    a = ap.arrays()
    for theta in (0.0, 0.5, 2.0):
        rc, Lo, dlo, dpo, _, _ = orc.dp_gain(P, ap, theta)
        Lm, dlm, sm, S0, why = sweep(a, N, prob.W, theta, 0.0)
        assert rc == 0 and why is None
        e = [np.abs(Lm - Lo).max() / np.abs(Lo).max(), np.abs(dlm - dlo).max() / np.abs(dlo).max(),
             np.abs(sm - dpo["s"]).max() / np.abs(dpo["s"]).max(), np.abs(S0 - dpo["S"][0]).max() / np.abs(dpo["S"][0]).max()]
        _, dpe = orc.dp_eval(P, ap, Lo * 0.9, dlo, theta, 1e-6)
        _, _, se, _, _ = sweep(a, N, prob.W, theta, 1e-6, L=Lo * 0.9, dl=dlo)
        e.append(abs(se[0] - dpe["s"][0]) / abs(dpe["s"][0]))
        assert max(e) < 1e-10, (theta, e)
    rc, *_ = orc.dp_gain(P, ap, 1e6)
    assert rc == 2 and sweep(a, N, prob.W, 1e6, 0.0)[4] == "M"          # isposdef(M) fails in both


def test_wide_step_model_against_the_gaussian_integral():
    prob, x0, l = ex.random_lq(14, 5, 8, 7)
    L = 0.05 * np.random.default_rng(0).standard_normal((8, 5, 14))
    P = orc.Problem(prob)
    _, xbar = orc.simulate_open(P, x0, l)
    _, ap = orc.approximate_model(P, l, xbar)
    th_max = ex.breakdown_theta(prob, x0, l, L, xbar)
    for theta in (0.0, 0.3 * th_max, 0.9 * th_max):
        exact, ok = ex.exact_value(prob, x0, l, None, L, xbar, theta)
        _, _, s, _, why = sweep(ap.arrays(), 8, prob.W, theta, 0.0, L=L)
        assert ok and why is None and abs(s[0] - exact) <= 1e-10 * abs(exact), (theta, s[0], exact)
