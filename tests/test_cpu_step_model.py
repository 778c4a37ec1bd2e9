"""The NumPy model of the augmented 16x16 Riccati step (tests/step_model.py: the algebra sweep_body implements with 14 + 6 MFMAs)
reproduces the oracle's gain sweep and policy evaluation: the design of the kernel is checked on the CPU, without a GPU."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc
from step_model import sweep


@pytest.mark.parametrize("n,m,N,kappa", [(12, 4, 50, 0.02), (4, 2, 20, 0.0), (2, 2, 10, 0.0)])
def test_augmented_step_model_matches_the_oracle(n, m, N, kappa):
    prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=3, kappa=kappa)
    P = orc.Problem(prob)
    u = 0.1 * np.random.default_rng(1).standard_normal((N, m))
    _, x = orc.simulate_open(P, x0, u)
    _, ap = orc.approximate_model(P, u, x)
    a = ap.arrays()
    for theta in (0.0, 2.0, 6.0):
        _, Lo, dlo, dpo, _, _ = orc.dp_gain(P, ap, theta)
        Lm, dlm, sm, _ = sweep(a, n, m, N, prob.W(0), theta, 0.0)
        e1 = np.abs(Lm - Lo).max() / np.abs(Lo).max()
        e2 = np.abs(dlm - dlo).max() / np.abs(dlo).max()
        e3 = np.abs(sm - dpo["s"]).max() / np.abs(dpo["s"]).max()
        _, dpe = orc.dp_eval(P, ap, Lo * 0.9, None, theta, 1e-6)
        _, _, se, _ = sweep(a, n, m, N, prob.W(0), theta, 1e-6, L=Lo * 0.9)
        e4 = abs(se[0] - dpe["s"][0]) / abs(dpe["s"][0])
        assert max(e1, e2, e3, e4) < 1e-10, (theta, e1, e2, e3, e4)
