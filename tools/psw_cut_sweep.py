"""The cut model of the time-parallel sweeps (switches psw_hop, psw_hop_e, psw_comp; hundredths of an ordinary step) swept around its defaults on
the latency kernel with two workgroups per sample: kernel time of a 128-sample batch by HIP events.   python tools/psw_cut_sweep.py [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prob, x0, u = rat.synthetic_lq_problem()
theta = np.abs(1.0 + 2.0 * np.random.default_rng(B).standard_normal(B))


def run(**sw):
    ctx = rat.Context(prob, max_batch=B)
    for k, v in sw.items():
        ctx.debug_set(k, v)
    ctx.set_initial(x0, u)
    for _ in range(5):
        out = ctx.solve_batch(x0, u, theta)
    ctx.profile(True)
    ms = []
    for _ in range(40):
        ctx.profile_reset(); out = ctx.solve_batch(x0, u, theta); ms.append(ctx.profile_get()["solve_block"]["ms"])
    return float(np.median(ms)), float(np.min(ms)), out


base = run()
print(f"defaults (hop 120, hop_e 140, comp 125): {base[0]:.4f} ms (min {base[1]:.4f})", flush=True)
for hop in (80, 100, 120, 150, 200, 300):
    for comp in (110, 125, 150):
        m, mn, out = run(psw_hop=hop, psw_comp=comp)
        same = all(np.array_equal(a, b) for a, b in zip(out[1:], base[2][1:]))
        print(f"psw_hop {hop} psw_comp {comp}: {m:.4f} ms (min {mn:.4f}) counts_same {same}", flush=True)
for hop_e in (80, 100, 120, 140, 180, 250):
    m, mn, out = run(psw_hop_e=hop_e)
    print(f"psw_hop_e {hop_e}: {m:.4f} ms (min {mn:.4f})", flush=True)
