"""Phase timeline of solve_block_kernel with E speculative candidates (wave 0 = candidate 0, last wave = gain sweeps) for the first 8
samples of a batch (build: make -C ratilqr.jl_amd/csrc diagp).  python tools/gpu_phases_block_e.py [B] [E]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diagp.so")
os.environ["RATILQR_BLOCK"] = "1"
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prob, x0, u = rat.synthetic_lq_problem()
ctx = rat.Context(prob, max_batch=B, spec_eps=E)
lib = rat.native.lib()
lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
for th in (1.0,):
    for _ in range(3):
        r = ctx.solve_batch(x0, u, np.full(B, th))
    print("path", ctx.get_path(B), "iters", np.unique(r[2]), "ls", np.unique(r[3]))
    out = np.zeros(640)
    lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 1024, 640)
    t = out.reshape(8, 2, 40)
    n = int(np.max(np.nonzero(t[0, 0])[0])) + 1
    arrive, leave = t[:, :, 0:n:2].mean(0), t[:, :, 1:n:2].mean(0)          # marks come in (before barrier, after barrier) pairs
    print(f"B = {B}, E = {E}, theta = {th}: total {t[:, 0, n - 1].mean():.0f} cycles")
    prev = np.zeros(2)
    for i in range(arrive.shape[1]):
        print(f"   phase {i:2d}   candidate-0 wave busy {arrive[0, i] - prev[0]:8.0f}   gain wave busy {arrive[1, i] - prev[1]:8.0f}   barrier released at {leave[0, i]:8.0f}")
        prev = leave[:, i]
