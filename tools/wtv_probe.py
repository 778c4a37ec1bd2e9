import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc
rng = np.random.default_rng(12)
n, m, N = 9, 3, 41
Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
A, B, x0 = 0.85 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
def spd(k, scale):
    G = rng.standard_normal((k, k))
    return scale * (np.eye(k) + 0.2 * G @ G.T / k)
for const in (True, False):
    W0 = spd(n, 1e-3)
    W = np.stack([W0 if const else spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=spd(n, 1.0), R=spd(m, 0.2), N=N, W=W, Qf=spd(n, 1.0), kappa=0.0)
    u = 0.1 * rng.standard_normal((N, m))
    theta = np.array([0.0, 0.5, 2.0, 6.0])
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=4)
    for psw, duo in ((0, 0), (1, 0), (1, 1)):
        ctx = rat.Context(prob, max_batch=4)
        ctx.debug_set("block_psw", psw); ctx.debug_set("psw_duo", duo)
        v, s, i, l = ctx.solve_batch(x0, u, theta)
        print("const" if const else "tv", "psw", psw, "duo", duo, "status", s.tolist(), so.tolist(), "iters", i.tolist(), io.tolist(), "relerr", np.abs(v / vo - 1).max())
