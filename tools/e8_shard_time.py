"""ms per batch of an E-candidate shard through the user-facing batch call (host buffers):  python tools/e8_shard_time.py [B] [E]"""
import os, sys, time
os.environ.setdefault("RATILQR_SPEC_FORCE", "1")     # handles of width E > 1 run the speculative kernels here (spec_eps is otherwise an upper bound)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prob, x0, u = rat.synthetic_lq_problem()
th = np.abs(1 + 2 * np.random.default_rng(1).standard_normal(B)) + 0.01
for blk in ("1", "0"):
    os.environ["RATILQR_BLOCK"] = blk
    ctx = rat.Context(prob, max_batch=B, spec_eps=E)
    for _ in range(5):
        r = ctx.solve_batch(x0, u, th)
    t0 = time.perf_counter()
    for _ in range(50):
        r = ctx.solve_batch(x0, u, th)
    dt = (time.perf_counter() - t0) / 50
    print(f"B = {B} E = {E} block = {blk}: path {ctx.get_path(B)}  {dt * 1e3:.3f} ms per batch; iters {np.unique(r[2])} ls {np.unique(r[3])}")
