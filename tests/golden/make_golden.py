"""Regenerates tests/golden/golden.json with the CPU oracle (oracle/libratilqr_oracle.so).

The reference itself is Julia and cannot run here (no julia binary), so these vectors are NOT reference
output: they are oracle output, and the oracle is pinned by the reference's known-answer tests
(tests/test_oracle_*.py).  Inputs are fully described by the seeds/parameters stored next to each vector.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ratilqr.jl_amd as rat  # noqa: E402
from oracle import oracle as orc  # noqa: E402

out = {}

# (a) ileqg_test.jl LQ problem: f = x + u, c = 0.5x'x + u'u, h = 0.5x'x, W = I, N = 10
I2 = np.eye(2)
prob = rat.LQRiskSensitiveProblem(I2, I2, Q=I2, R=2 * I2, N=10, W=I2, Qf=I2)
P = orc.Problem(prob)
u = np.ones((10, 2))
_, x = orc.simulate_open(P, np.zeros(2), u)
_, ap = orc.approximate_model(P, u, x)
ga = {}
for th in (0.0, 1e-8, 0.05):
    _, L, dl, dp, _, _ = orc.dp_gain(P, ap, th)
    ga[repr(th)] = dict(s=dp["s"].tolist(), L=L.tolist(), dl=dl.tolist(), S=dp["S"].tolist(), sv=dp["sv"].tolist())
out["ileqg_test_lq"] = dict(x=x.tolist(), gain=ga)

# (b) nonlinear test problem (ileqg_test.jl:151-161, ce_test.jl:14-24)
pl = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2))
PP = orc.Problem(pl)
u = 0.1 * np.ones((10, 2))
nl = {}
for th in (0.0, 0.1, 0.3, 0.43, 0.5):
    s = orc.ILEQGSolver(PP)
    rc = s.solve(np.zeros(2), u, th)
    nl[repr(th)] = dict(rc=rc, value=s.s.value_current, iters=s.s.iter_current, ls=s.s.n_ls_evals,
                        hist=s.eps_history.tolist(), x=s.x_array.tolist(), l=s.l_array.tolist(), L=s.L_array.tolist())
out["nonlinear_test"] = nl

# (c) BASELINE config 2: 256 theta on the synthetic LQ problem (N=50, n=12, m=4, seed 0)
prob, x0, u0 = rat.synthetic_lq_problem()
P = orc.Problem(prob)
lo, hi = 1.0, 64.0          # bisection for the breakdown theta of initialize!
for _ in range(50):
    mid = 0.5 * (lo + hi)
    v, st, _, _ = orc.compute_value_batch(P, x0, u0, [mid])
    if st[0] == 1:
        hi = mid
    else:
        lo = mid
theta_max = lo
theta = np.linspace(0.01, 0.8 * theta_max, 256)
v, st, it, ls = orc.compute_value_batch(P, x0, u0, theta, nthreads=8)
out["config2"] = dict(theta_breakdown=theta_max, theta=theta.tolist(), value=v.tolist(), status=st.tolist(),
                      iters=it.tolist(), ls=ls.tolist())
th_edge = np.array([0.0, theta_max * 0.999, theta_max * 1.001, 2 * theta_max])
v, st, it, ls = orc.compute_value_batch(P, x0, u0, th_edge)
out["config2_edge"] = dict(theta=th_edge.tolist(), value=[None if not np.isfinite(a) else a for a in v], status=st.tolist())

# (d) CE solve with an injected N(0,1) stream (seed 2024) on the same problem
z = np.random.default_rng(2024).standard_normal(20000)
ce = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=64, num_elite=8, nthreads=8)
rc, th_opt, xx, ll, LL, val, tmin, tmax = ce.solve(P, x0, u0, 0.1)
out["ce_config"] = dict(rc=rc, z_seed=2024, num_samples=64, num_elite=8, kl_bound=0.1, theta_opt=th_opt, value=val,
                        theta_min=tmin, theta_max=tmax, mu=ce.c.mu, sigma=ce.c.sigma, mu_init=ce.c.mu_init,
                        sigma_init=ce.c.sigma_init, zpos=ce.c.zpos, n_solves=ce.c.n_solves,
                        l0=ll[0].tolist(), L0=LL[0].tolist())

# (e) BASELINE config 1 plumbing instance: N=20, n=4, m=2 LQ
prob4, x04, u04 = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)
P4 = orc.Problem(prob4)
th4 = np.array([0.0, 0.5, 2.0, 8.0])
v, st, it, ls = orc.compute_value_batch(P4, x04, u04, th4)
s4 = orc.ILEQGSolver(P4)
s4.solve(x04, u04, 0.5)
out["config1_n4"] = dict(theta=th4.tolist(), value=[None if not np.isfinite(a) else a for a in v], status=st.tolist(),
                         iters=it.tolist(), L0=s4.L_array[0].tolist(), x_end=s4.x_array[-1].tolist())

# (f) cubic-drift problem with backtracking line search (seed 5, kappa 0.05)
probk, x0k, u0k = rat.synthetic_lq_problem(seed=5, kappa=0.05)
Pk = orc.Problem(probk)
thk = np.array([0.0, 2.0, 5.0, 6.0])
v, st, it, ls = orc.compute_value_batch(Pk, x0k, u0k, thk)
sk = orc.ILEQGSolver(Pk)
sk.solve(x0k, u0k, 5.0)
out["cubic_backtracking"] = dict(theta=thk.tolist(), value=v.tolist(), status=st.tolist(), iters=it.tolist(), ls=ls.tolist(),
                                 hist_theta5=sk.eps_history.tolist())

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote golden.json; theta_breakdown =", theta_max)
