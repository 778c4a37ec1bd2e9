"""Where one PETS compute_cost call of 10k trajectories spends its time: Python wrapper, raw C-ABI call, kernels (profiles/r04_rocprof_summary.md)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import pets, _native as nv

rp = np.random.default_rng(8)
Ap = 0.9 * np.linalg.qr(rp.standard_normal((12, 12)))[0]
Bp = rp.standard_normal((12, 4)) / np.sqrt(12)
gprob = rat.LQGenerativeProblem(Ap, Bp, 30, ("gaussian", np.zeros(12), 0.03 * np.eye(12)), Q=np.eye(12), R=0.1 * np.eye(4), Qf=np.eye(12), kappa=-0.01)
xp0 = rp.standard_normal(12)
for S_, K_ in ((100, 100), (1000, 100), (1000, 1000)):
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([np.eye(4)] * 30), num_control_samples=S_, num_trajectory_samples=K_)
    ctrl = nv.f64(0.3 * rp.standard_normal((S_, 30, 4)))
    ctx = ds.context(gprob)
    for w16 in (1, 2, 3, 0):
        ctx.debug_set("pets_wave16", w16)
        for _ in range(5):
            pets.compute_cost_serial(ds, gprob, xp0, ctrl, None, False, seed=11)
        reps = 200 if S_ * K_ <= 100000 else 20
        t0 = time.perf_counter()
        for i in range(reps):
            pets.compute_cost_serial(ds, gprob, xp0, ctrl, None, False, seed=11 + i)
        t_wrap = (time.perf_counter() - t0) / reps
        cost = np.zeros(S_)
        x0 = nv.f64(xp0)
        args = (ctx.h, nv.P(x0), nv.P(ctrl), C.c_int64(S_), C.c_int64(K_), 0, None, None, C.c_uint64(5), nv.P(cost))
        f = nv.lib().rat_pets_compute_cost
        t0 = time.perf_counter()
        for i in range(reps):
            f(*args)
        t_raw = (time.perf_counter() - t0) / reps
        ctx.profile(True, kinds=["pets"])
        ctx.profile_reset()
        for i in range(reps):
            f(*args)
        pk = ctx.profile_get()["pets"]
        ctx.profile(False)
        print(f"S={S_} K={K_} wave16={w16}: wrapper {t_wrap * 1e6:.1f} us  raw C call {t_raw * 1e6:.1f} us  kernels {pk['ms'] / pk['launches'] * 1e3:.1f} us"
              f"  -> {S_ * K_ / t_wrap / 1e6:.1f} M traj/s (wrapper)  {S_ * K_ / t_raw / 1e6:.1f} M (raw)")
