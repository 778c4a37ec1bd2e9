"""BASELINE config 4 under the profiler (tools/profile_r03_aux.sh): fresh NelderMeadBilevelOptimizationSolver solves on the headline problem."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import nelder_mead as nm
prob, x0, u0 = rat.synthetic_lq_problem()
nms = rat.NelderMeadBilevelOptimizationSolver(device=0)
t = []
for _ in range(12):
    nms.c.has_c_high = nms.c.has_c_low = 0
    nms.c.theta_high_init, nms.c.theta_low_init = 3.0, 1e-8
    nb0 = int(nms.c.n_batches)
    t0 = time.perf_counter(); r = nm.solve_(nms, prob, x0, u0, 0.1); t.append(time.perf_counter() - t0)
print(json.dumps({"ms_per_solve": min(t) * 1e3, "batched_device_calls": int(nms.c.n_batches) - nb0, "theta_opt": r[0]}))
