"""solve_block_kernel against solve_block_psw_kernel (time-parallel sweeps; switch block_psw) on shards of the headline CE batch:
kernel time per batch from the library's HIP events, median of R batches.   python tools/psw_block_time.py [B ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat

Bs = [int(a) for a in sys.argv[1:]] or [1, 128, 256]
R = int(os.environ.get("PSW_REPS", 30))
prob, x0, u = rat.synthetic_lq_problem()
rng = np.random.default_rng(1000)
out = {}
for B in Bs:
    theta = np.abs(1.0 + 2.0 * rng.standard_normal(B)) + 1e-3
    row = {}
    res = {}
    for psw in (0, 1):
        ctx = rat.Context(prob, max_batch=B, spec_eps=1)
        ctx.debug_set("block_psw", psw)
        for k in ("psw_hop", "psw_hop_e", "psw_comp"):
            if os.environ.get(k.upper()):
                ctx.debug_set(k, int(os.environ[k.upper()]))
        for _ in range(5):
            res[psw] = ctx.solve_batch(x0, u, theta)
        ts = []
        for _ in range(R):
            ctx.profile(True); ctx.profile_reset()
            ctx.solve_batch(x0, u, theta)
            p = ctx.profile_get()
            ctx.profile(False)
            ts.append(p["solve_block"]["ms"])
        row["psw" if psw else "block"] = {"median_ms": float(np.median(ts)), "min_ms": float(np.min(ts))}
    (v0, s0, i0, l0), (v1, s1, i1, l1) = res[0], res[1]
    fin = np.isfinite(v0)
    row["same_counts"] = bool(np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1))
    row["value_err"] = float(np.abs(v1[fin] - v0[fin]).max() / np.abs(v0[fin]).max())
    row["speedup"] = row["block"]["median_ms"] / row["psw"]["median_ms"]
    out[str(B)] = row
    print(f"B={B}: block {row['block']['median_ms']:.4f} ms (min {row['block']['min_ms']:.4f})  psw {row['psw']['median_ms']:.4f} ms (min {row['psw']['min_ms']:.4f})"
          f"  x{row['speedup']:.2f}  same_counts {row['same_counts']}  value_err {row['value_err']:.1e}", flush=True)
print(json.dumps(out))
