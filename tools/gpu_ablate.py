"""Kernel-time split: theta = 0 (no inverse, 8 MFMA / step) vs theta > 0 (inverse + 14 MFMA / step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
prob, x0, u = rat.synthetic_lq_problem()
for E in (1, 8):
    ctx = rat.Context(prob, max_batch=1024, spec_eps=E)
    for name, th in (("theta=0", np.zeros(1024)), ("theta=1", np.ones(1024))):
        for _ in range(3): ctx.solve_batch(x0, u, th)
        ctx.profile(True); ctx.profile_reset()
        for _ in range(10): ctx.solve_batch(x0, u, th)
        p = ctx.profile_get(); ctx.profile(False)
        print(E, name, {k: round(v["ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in p.items()}, "us/launch")
