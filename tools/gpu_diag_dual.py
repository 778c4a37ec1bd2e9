"""Per-segment cycles of one paired sweep step inside solve_fused_kernel (diagnostic build: make -C ratilqr.jl_amd/csrc diag; the stamps
serialise the segments, so the sum exceeds the undisturbed step -- the split is what it is for)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
prob, x0, u = rat.synthetic_lq_problem()
ctx = rat.Context(prob, max_batch=1024)
names = ["top / M build", "elimination (12 rounds)", "Y', T MFMAs + racc", "F MFMAs + exchange writes", "exchange reads + 4x4 solve + ua", "fx + final MFMAs"]
lib = rat.native.lib()
lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
for th in (1.0,):
    for _ in range(3):
        ctx.solve_batch(x0, u, np.full(1024, th))
    out = np.zeros(64)
    lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 128, 64)
    d = out.reshape(8, 8)[:, :6].mean(0) / 50.0
    print(f"theta={th}: last paired sweep of the solve, cycles per step: " + ", ".join(f"{n} = {c:.0f}" for n, c in zip(names, d)) + f" | total {d.sum():.0f}")
