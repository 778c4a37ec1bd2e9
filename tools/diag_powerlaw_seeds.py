import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc
np.set_printoptions(precision=17, linewidth=200)
for seed in (236, 386, 584, 704):
    rng = np.random.default_rng(9000 + seed)
    n, N = int(rng.integers(1, 5)), int(rng.integers(2, 31))
    a, b, p = float(rng.choice([1.0, 1.3, 2.0, 0.8])), float(rng.choice([1.0, 1.5, 2.0])), float(rng.choice([2.0, 2.5, 3.0, 4.0]))
    W = (10.0 ** rng.uniform(-3, -1)) * np.eye(n)
    prob = rat.PowerLawRiskSensitiveProblem(n, N, W, a=a, b=b, p=p, cx=float(rng.uniform(0.5, 2)), cu=float(rng.uniform(0.5, 2)))
    x0 = rng.uniform(0.0, 0.6, n) * (1 if rng.random() < 0.85 else -1)
    u = rng.uniform(0.02, 0.3, (N, n))
    theta = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-2, 1.5, 7))])
    E = int(rng.choice([1, 1, 2, 4, 8]))
    P = orc.Problem(prob)
    vo, so, io, lo = orc.compute_value_batch(P, x0, u, theta, nthreads=4)
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=E)
    vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
    print("seed", seed, (n, N, a, b, p), "E", E, "x0", x0)
    print(" vo", vo); print(" vg", vg); print(" so", so, "sg", sg); print(" io", io, "ig", ig)
    for k in range(theta.size):
        if so[k] == sg[k] and io[k] == ig[k] and (not np.isfinite(vo[k]) or abs(vg[k] - vo[k]) <= 1e-9 * abs(vo[k])):
            continue
        S = orc.ILEQGSolver(P)
        rc = S.solve(x0, u, float(theta[k]))
        rg = rat.Context(prob, spec_eps=1).solve(x0, u, float(theta[k]))
        eo, eg = S.eps_history, np.asarray(rg["eps_history"])
        print("  theta", theta[k], "oracle rc", rc, "iters", S.s.iter_current if hasattr(S.s, "iter_current") else None, "device status", rg["status"], "iters", rg["iters"])
        m = min(len(eo), len(eg))
        d = [i for i in range(m) if eo[i][0] != eg[i][0]]
        print("  eps history lengths", len(eo), len(eg), "first differing eps index", d[:1], "value diffs at the first 5 entries",
              [(float(eo[i][1]), float(eg[i][1])) for i in range(min(m, 5))])
        if d:
            i0 = d[0]
            print("  around the first difference: oracle", eo[max(0, i0 - 2): i0 + 2].tolist(), "device", eg[max(0, i0 - 2): i0 + 2].tolist())
        print("  min over l: oracle", float(np.min(S.l_array)), "device", float(np.min(rg["l"])), "| min over x: oracle", float(np.min(S.x_array)), "device", float(np.min(rg["x"])))
