"""One PETS solve! (pets.jl:270-281; BASELINE config 5 shape: 100 control samples x 100 rollouts, N = 30, 5 iterations) with the loop
over control sequences on the device (switch pets_device = 1) and on the host (0): wall time per solve and the kernel sum from the
library's HIP events.   python tools/pets_solve_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import pets

r = np.random.default_rng(8)
n, m, Nh = 12, 4, 30
A = 0.9 * np.linalg.qr(r.standard_normal((n, n)))[0]
cov = 0.02 * np.eye(n) + 0.01 * np.outer(np.ones(n), np.ones(n)) / n
prob = rat.LQGenerativeProblem(A, r.standard_normal((n, m)) / np.sqrt(n), Nh, ("gaussian", 0.05 * r.standard_normal(n), cov), Q=np.eye(n), R=0.1 * np.eye(m),
                               Qf=2 * np.eye(n), kappa=0.0, l1u=0.2)
x0 = r.standard_normal(n)
for S, K in ((100, 100), (1000, 1000)):
    res = {}
    for dev in (0, 1):
        ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, m)), np.stack([0.3 * np.eye(m)] * Nh), num_control_samples=S, num_trajectory_samples=K,
                                                      num_elite=max(2, S // 10), iter_max=5)
        ctx = ds.context(prob)
        ctx.debug_set("pets_device", dev)
        for _ in range(3):
            pets.solve_(ds, prob, x0, np.random.default_rng(5), seed=11)
        import ctypes as C
        from ratilqr.jl_amd import _native as nv
        zc = nv.f64(np.random.default_rng(5).standard_normal(5 * S * Nh * m))
        ts = []
        for _ in range(20):                                # the library call alone (control normals drawn beforehand)
            t0 = time.perf_counter()
            nv.check(nv.lib().rat_pets_solve(ctx.h, C.byref(ds.c), nv.P(nv.f64(x0)), 0, nv.P(zc), None, None, C.c_uint64(11)))
            ts.append(time.perf_counter() - t0)
        ctx.profile(True); ctx.profile_reset()
        mu, _ = pets.solve_(ds, prob, x0, np.random.default_rng(5), seed=11)
        p = ctx.profile_get(); ctx.profile(False)
        ksum = sum(v["ms"] for v in p.values())
        res[dev] = (float(np.median(ts)) * 1e3, ksum, mu.copy())
        print(f"S={S} K={K} pets_device={dev}: solve! {res[dev][0]:.3f} ms (rat_pets_solve alone, median of 20), kernels {ksum:.3f} ms "
              f"({ {k: round(v['ms'], 3) for k, v in p.items() if v['launches']} })", flush=True)
    print("   mu bit-identical:", bool(np.array_equal(res[0][2], res[1][2])))
