"""Result dump of the LARGE-batch kernels for tools/novf_diff.sh (the random problems of soak_parity.py are 12-sample batches: the workgroup
kernels): the headline batch and its cubic-drift variant at 300 / 1024 / 4100 samples (two waves per sample, one persistent wave per sample,
two per SIMD), a W(k) problem, one CE solve and one Nelder-Mead solve.   python tools/novf_batch.py out.npz"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat

out = {}
for kappa in (0.0, 0.05):
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
    for B in (300, 1024, 4100):
        theta = np.abs(1.0 + 2.0 * np.random.default_rng(B).standard_normal(B)); theta[::97] = 0.0; theta[5] = 80.0
        ctx = rat.Context(prob, max_batch=B)
        v, s, i, l = ctx.solve_batch(x0, u, theta)
        out[f"v_{kappa}_{B}"], out[f"s_{kappa}_{B}"], out[f"i_{kappa}_{B}"], out[f"l_{kappa}_{B}"] = v, s, i, l
prob, x0, u = rat.synthetic_lq_problem()
r = rat.cross_entropy.solve_(rat.CrossEntropyBilevelOptimizationSolver(), prob, x0, u, np.random.default_rng(3), 0.1)
out["ce_theta_value"], out["ce_x"], out["ce_l"], out["ce_L"] = np.array([r[0], r[4], r[5], r[6]]), r[1], r[2], r[3]
r = rat.nelder_mead.solve_(rat.NelderMeadBilevelOptimizationSolver(), prob, x0, u, 0.1)
out["nm_theta_value"], out["nm_x"], out["nm_l"], out["nm_L"] = np.array([r[0], r[4]]), r[1], r[2], r[3]
rng = np.random.default_rng(9)                    # time-varying noise covariance: the W(k) instantiations of the latency kernel
n, m, N = 12, 4, 40
G = rng.standard_normal((N, n, n))
Wk = 1e-3 * (np.eye(n)[None] + 0.2 * G @ np.transpose(G, (0, 2, 1)) / n)
A = 0.9 * np.linalg.qr(rng.standard_normal((n, n)))[0]
pw = rat.LQRiskSensitiveProblem(A, rng.standard_normal((n, m)) / np.sqrt(n), Q=np.eye(n), R=0.1 * np.eye(m), N=N, W=Wk, Qf=np.eye(n))
theta = np.linspace(0.0, 8.0, 40)
v, s_, i_, l_ = rat.Context(pw, max_batch=40).solve_batch(rng.standard_normal(n), np.zeros((N, m)), theta)
out["v_wk"], out["s_wk"], out["i_wk"], out["l_wk"] = v, s_, i_, l_
np.savez(sys.argv[1], **out)
print("dumped", len(out), "arrays")
