"""BASELINE config 5: PETS forward-simulation throughput -- stochastic rollouts (N = 30, n = 12, m = 4, cubic drift, Gaussian process
noise from the device Philox generator) per second through rat_pets_compute_cost (pets.jl:128-157), host buffers in and out.
python tools/pets_bench.py  (on an MI355X);  rocprofv3 --kernel-trace --stats -- python3 tools/pets_bench.py  for the kernel time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import pets

n, m, Nh = 12, 4, 30
r = np.random.default_rng(8)
A = 0.9 * np.linalg.qr(r.standard_normal((n, n)))[0]
B = r.standard_normal((n, m)) / np.sqrt(n)
prob = rat.LQGenerativeProblem(A, B, Nh, ("gaussian", np.zeros(n), 0.03 * np.eye(n)), Q=np.eye(n), R=0.1 * np.eye(m), Qf=np.eye(n), kappa=-0.01)
x0 = r.standard_normal(n)
out = []
for S, K in ((100, 100), (1000, 100), (1000, 1000)):
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, m)), np.stack([np.eye(m)] * Nh), num_control_samples=S, num_trajectory_samples=K)
    ctrl = 0.3 * r.standard_normal((S, Nh, m))
    for _ in range(3):
        c = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11)
    reps = 20
    t0 = time.perf_counter()
    for i in range(reps):
        c = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11 + i)
    dt = (time.perf_counter() - t0) / reps
    assert np.all(np.isfinite(c))
    out.append({"control_samples": S, "trajectories_per_sample": K, "trajectories": S * K, "ms_per_call": dt * 1e3,
                "trajectories_per_s": S * K / dt, "steps_per_s": S * K * Nh / dt})
print(json.dumps({"metric": "PETS stochastic rollouts/s (N=30, n=12, m=4), host buffers in/out", "runs": out}))
