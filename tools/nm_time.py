"""RAT iLQR++ (Nelder-Mead) solve time by speculation depth (switch nm_depth):  python tools/nm_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import nelder_mead as nm
prob, x0, u0 = rat.synthetic_lq_problem()
ref = None
for depth in (3, 2, 1, 0):
    nms = rat.NelderMeadBilevelOptimizationSolver()
    ctx = nms.context(prob)
    ctx.debug_set("nm_depth", depth)
    def fresh():
        nms.c.has_c_high = 0; nms.c.has_c_low = 0
        nb0, ns0 = int(nms.c.n_batches), int(nms.c.n_solves)
        r = nm.solve_(nms, prob, x0, u0, 0.1)
        return r, int(nms.c.n_batches) - nb0, int(nms.c.n_solves) - ns0
    for _ in range(3):
        r, nb, ns = fresh()
    t0 = time.perf_counter()
    for _ in range(30):
        r, nb, ns = fresh()
    dt = (time.perf_counter() - t0) / 30
    key = (r[0], r[4], nms.c.iter_current, ns) + tuple(np.asarray(r[k]).tobytes() for k in (1, 2, 3))
    same = ref is None or key == ref
    ref = ref or key
    print(f"nm_depth {depth}: {dt * 1e3:.3f} ms per solve; iterations {nms.c.iter_current}, sequential evaluations {ns}, device calls {nb}; theta_opt {r[0]:.6g} "
          f"objective {r[4]:.9g}; identical to depth 3: {same}")
