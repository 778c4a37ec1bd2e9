"""Per-segment cycle shares of one sweep wave (diagnostic build: make -C ratilqr.jl_amd/csrc diag)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diag.so")
os.environ["RATILQR_FUSED"] = "0"      # per-phase kernels (the stamps sit in their bodies)
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
prob, x0, u = rat.synthetic_lq_problem()
names = ["top/M build", "elimination", "Y,V~ MFMA", "T,F MFMA", "exchange+solve", "Fx + final MFMA"]
for E in (1, 8):
    ctx = rat.Context(prob, max_batch=1024, spec_eps=E)
    for th in (0.0, 1.0):
        ctx.solve_batch(x0, u, np.full(1024, th))
        out = np.zeros(128)
        rat.native.lib().rat_diag_read(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        d = out[:64].reshape(8, 8)[:, :6].mean(0) / 50.0
        dr = out[64:].reshape(8, 8)[:, :3].mean(0) / 50.0       # last launch = line-search eval sweep, per time step
        pro, whole = out[64:].reshape(8, 8)[:, 3].mean(), out[64:].reshape(8, 8)[:, 4].mean()
        print(f"E={E} theta={th}: cycles/step " + ", ".join(f"{n}={c:.0f}" for n, c in zip(names, d)) + f" | total {d.sum():.0f}")
        print(f"      rollin cycles/step: dx+u={dr[0]:.0f}, xu+x'={dr[1]:.0f}, tile+rest={dr[2]:.0f} | total {dr.sum():.0f}"
              f" | wave: prologue {pro:.0f} cycles, entry..loop end {whole:.0f} cycles, gaps {out[64:].reshape(8, 8)[:, 5].mean():.0f}")
