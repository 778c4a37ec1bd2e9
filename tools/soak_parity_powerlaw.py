"""Randomised parity soak of the power-law family (f = x.^a + u.^b, c = cx sum x.^p + cu sum u.^p; n = m <= 4) against the CPU oracle:
random exponents, horizons, noise scales, speculation widths, theta from 0 through infeasible; DomainErrors included.
Round 1, 600 problems (4,800 solves; oracle statuses: 985 ok, 409 iter_max, 1,932 DomainError, 1,474 M not PD): statuses always equal;
4 problems differ in the iteration at which a DomainError is met (a control that is analytically 0 takes either sign);
4 problems with 77-90 iterations or 26-51 line-search evaluations per solve agree in every count but in value only to 2e-9 .. 5e-8
(device pow() and libm pow() differ in the last place and a long nonconvex descent amplifies it).
  SOAK_N=600 python tools/soak_parity_powerlaw.py   (on an MI355X; ~15 s)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc
bad = 0
stat = {}
for seed in range(int(os.environ.get("SOAK_N", "600"))):
    rng = np.random.default_rng(9000 + seed)
    n, N = int(rng.integers(1, 5)), int(rng.integers(2, 31))
    a, b, p = float(rng.choice([1.0, 1.3, 2.0, 0.8])), float(rng.choice([1.0, 1.5, 2.0])), float(rng.choice([2.0, 2.5, 3.0, 4.0]))
    W = (10.0 ** rng.uniform(-3, -1)) * np.eye(n)
    prob = rat.PowerLawRiskSensitiveProblem(n, N, W, a=a, b=b, p=p, cx=float(rng.uniform(0.5, 2)), cu=float(rng.uniform(0.5, 2)))
    x0 = rng.uniform(0.0, 0.6, n) * (1 if rng.random() < 0.85 else -1)
    u = rng.uniform(0.02, 0.3, (N, n))
    theta = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-2, 1.5, 7))])
    E = int(rng.choice([1, 1, 2, 4, 8]))
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=16)
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=E)
    vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
    fin = np.isfinite(vo)
    ok = np.array_equal(sg, so) and np.array_equal(ig, io) and np.array_equal(lg, lo) and np.array_equal(fin, np.isfinite(vg)) \
        and (not fin.any() or np.all(np.abs(vg[fin] - vo[fin]) <= 1e-9 * np.abs(vo[fin])))
    for s_ in so: stat[int(s_)] = stat.get(int(s_), 0) + 1
    if not ok:
        bad += 1
        print("values", vo[fin].tolist(), (np.abs(vg[fin]-vo[fin])/np.abs(vo[fin])).tolist())
        print("MISMATCH seed", seed, (n, N, a, b, p), E, so.tolist(), sg.tolist(), io.tolist(), ig.tolist(), lo.tolist(), lg.tolist())
print("power-law soak:", bad, "mismatches; oracle status histogram", stat)
