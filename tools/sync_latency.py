"""Host-side wait latency of a synchronous library call (rat_ileqg_solve: one sample, x / l / L returned) under the HIP scheduling policies:
   python tools/sync_latency.py [auto|spin|yield|blocking]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
hip = ctypes.CDLL("libamdhip64.so")
flags = {"auto": 0, "spin": 1, "yield": 2, "blocking": 4}[mode]
if mode != "auto":
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(flags)))
import numpy as np
import ratilqr.jl_amd as rat
prob, x0, u = rat.synthetic_lq_problem()
ctx = rat.Context(prob)
for _ in range(20):
    ctx.solve(x0, u, 1.0)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); ctx.solve(x0, u, 1.0); ts.append(time.perf_counter() - t0)
print(f"{mode}: rat_ileqg_solve wall median {np.median(ts) * 1e3:.4f} ms  min {min(ts) * 1e3:.4f}")
th = np.abs(1 + 2 * np.random.default_rng(0).standard_normal(1024)) + 0.01
c2 = rat.Context(prob, max_batch=1024)
for _ in range(20):
    c2.solve_batch(x0, u, th)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); c2.solve_batch(x0, u, th); ts.append(time.perf_counter() - t0)
print(f"{mode}: rat_ileqg_solve_batch(1024, host buffers) wall median {np.median(ts) * 1e3:.4f} ms  min {min(ts) * 1e3:.4f}")
