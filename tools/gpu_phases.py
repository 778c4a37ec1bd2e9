"""Phase timeline of solve_fused_kernel (build: make -C ratilqr.jl_amd/csrc diagp): cycles between phase boundaries of
the first 8 samples of a 1024-sample batch, next to the in-loop cycle counts the per-step diagnostics give."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diagp.so")
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
prob, x0, u = rat.synthetic_lq_problem()
ctx = rat.Context(prob, max_batch=1024, spec_eps=1)
import sys as _sys
DUAL = os.environ.get("RATILQR_FUSED_DUAL", "1") != "0"
if DUAL:        # default: policy evaluation + following gain sweep paired in one pass
    names = ["rollin0", "fence", "init eval + gain 1", "fence+commit",
             "rollin 1", "fence", "eval 1 + gain 2", "fence+select", "rollin 2", "fence", "eval 2", "fence+select"]
else:
    names = ["rollin0", "fence", "sweep init", "fence",
             "gain 1", "fence", "rollin 1", "fence", "eval 1", "fence+select",
             "gain 2", "fence", "rollin 2", "fence", "eval 2", "fence+select"]
for th in (0.0, 1.0):
    for _ in range(2):
        ctx.solve_batch(x0, u, np.full(1024, th))
    out = np.zeros(320)
    lib = rat.native.lib()
    lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
    lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 256, 320)
    t = out.reshape(8, 40)[:, :len(names)]
    d = np.diff(np.concatenate([np.zeros((8, 1)), t], axis=1), axis=1).mean(0)
    print(f"theta={th}: total {t[:, -1].mean():.0f} cycles = {t[:, -1].mean() / 2.34e3:.1f} us at 2.34 GHz")
    print("   " + ", ".join(f"{n}={c:.0f}" for n, c in zip(names, d)))
    inner = np.zeros(256)
    lib.rat_diag_read_n(ctx.h, inner.ctypes.data_as(C.POINTER(C.c_double)), 640, 256)
    inner = inner.reshape(8, 32)
    lab = ["sweep gain", "sweep eval", "sweep init", "rollin open", "rollin closed", "dual init+gain", "dual eval+gain"]
    print("   last occurrence of each body: prologue / time loop / epilogue cycles")
    for q, nm in enumerate(lab):
        m = inner[:, 4 * q:4 * q + 4]
        if np.all(m[:, 0] == 0):
            continue
        dd = np.diff(m, axis=1).mean(0)
        print(f"     {nm:16s} {dd[0]:8.0f} / {dd[1]:8.0f} / {dd[2]:8.0f}")
    ex = inner[:, 28:31] - inner[:, 24:25]
    print("   dual eval+gain prologue marks (cycles after entry): scalars in", ex[:, 0].mean(), "terminal tile in", ex[:, 1].mean(), "setup done", ex[:, 2].mean())
