"""A workload whose line search really backtracks: the SURVEY 8(d) problem with cubic drift kappa and a risk
parameter near the feasibility boundary (kappa = 0.06: 5 iterations with 10 line-search evaluations at theta = 2, 13 at theta = 5), solved for a whole batch at speculation widths E = 1, 2, 4, 8.  Results are identical for
every E (App. B.17); what changes is how many line-search evaluations run concurrently instead of serially.
  python tools/backtrack_bench.py [--kappa 0.06] [--theta 2 5] [--batch 1024]   (on an MI355X)"""
import argparse
import os
os.environ.setdefault("RATILQR_SPEC_FORCE", "1")     # handles of width E > 1 run the speculative kernels here (spec_eps is otherwise an upper bound)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ratilqr.jl_amd as rat

ap = argparse.ArgumentParser()
ap.add_argument("--kappa", type=float, default=0.06)
ap.add_argument("--theta", type=float, nargs="*", default=[2.0, 5.0])
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--widths", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()

prob, x0, u = rat.synthetic_lq_problem(kappa=a.kappa)
B = a.batch
for th0 in a.theta:
    th = torch.full((B,), th0, dtype=torch.float64, device="cuda")
    ref = None
    for E in a.widths:
        ctx = rat.Context(prob, max_batch=B, spec_eps=E)
        ctx.set_initial(x0, u)
        v = torch.empty(B, dtype=torch.float64, device="cuda")
        st, it, ls = (torch.empty(B, dtype=torch.int32, device="cuda") for _ in range(3))
        for _ in range(3):
            ctx.solve_batch_dev(th.data_ptr(), B, v.data_ptr(), st.data_ptr(), it.data_ptr(), ls.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            ctx.solve_batch_dev(th.data_ptr(), B, v.data_ptr(), st.data_ptr(), it.data_ptr(), ls.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        out = (v[0].item(), int(st[0]), int(it[0]), int(ls[0]))
        same = ref is None or (out[0] == ref[0] and out[1:3] == ref[1:3])
        ref = ref or out
        print(f"kappa {a.kappa} theta {th0} B {B} E {E}: {ms:8.3f} ms/batch = {B / ms * 1e3 / 1e6:6.3f} M solves/s | value {out[0]:.12g} status {out[1]} "
              f"iters {out[2]} ls_evals {out[3]} | identical to E={a.widths[0]}: {same}", flush=True)
        del ctx
