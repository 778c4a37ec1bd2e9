"""Phase timeline of solve_block_psw_kernel (waves 0 and 1) for the first 8 samples of a batch: cycles at which the waves reach / leave
every workgroup barrier (build: make -C ratilqr.jl_amd/csrc diagp).  python tools/gpu_phases_bpsw.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diagp.so")
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prob, x0, u = rat.synthetic_lq_problem()
lib = rat.native.lib()
lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
for psw in (0, 1):
    ctx = rat.Context(prob, max_batch=B, spec_eps=1)
    ctx.debug_set("block_psw", psw)
    for k in ("psw_hop", "psw_hop_e", "psw_comp"):
        if os.environ.get(k.upper()):
            ctx.debug_set(k, int(os.environ[k.upper()]))
    names = ["init", "rollout 0", "init eval | gain 1", "commit", "rollout 1", "eval 1 | gain 2", "select", "rollout 2", "eval 2 | -", "select", "end"] if not psw else \
            ["init", "rollout 0", "init eval | gain 1", "commit", "rollout 1", "d_c", "eval 1 | gain 2", "select", "rollout 2", "d_c", "eval 2 (4 waves)", "select", "end"]
    for _ in range(3):
        ctx.solve_batch(x0, u, np.full(B, 1.0))
    out = np.zeros(640)
    lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 1024, 640)
    if psw:        # (round 6: this kernel's barrier marks are s_memrealtime stamps shared by its two workgroups: tools/gpu_phases_duo.py prints them)
        ctx.debug_set("psw_duo", 0)
        ctx.solve_batch(x0, u, np.full(B, 1.0))
        print(f"block_psw = 1, B = {B}: barrier timeline in tools/gpu_phases_duo.py; the sweeps' own phases (one workgroup per sample):")
    t = out.reshape(8, 2, 40)
    n = int(np.max(np.nonzero(t[0, 0])[0])) + 1 if not psw else 0
    arrive, leave = t[:, :, 0:n:2].mean(0), t[:, :, 1:n:2].mean(0)
    if not psw:
        print(f"block_psw = {psw}, B = {B}: total {t[:, 0, n - 1].mean():.0f} cycles")
    prev = np.zeros(2)
    for i in range(min(arrive.shape[1], leave.shape[1])):
        nm = names[i] if i < len(names) else f"phase {i}"
        print(f"   {nm:22s} wave 0 busy {arrive[0, i] - prev[0]:8.0f}   wave 1 busy {arrive[1, i] - prev[1]:8.0f}   barrier released at {leave[0, i]:8.0f}")
        prev = leave[:, i]
    if psw:
        out = np.zeros(4096)
        lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 4096, 4096)
        t = out.reshape(8, 8, 4, 16)
        for mode, nm, P in ((2, "initialize!'s evaluation (2 waves)", 2), (5, "first gain sweep (2 waves)", 2), (4, "speculative gain sweep (2 waves)", 2), (1, "last evaluation (4 waves)", 4)):
            tt = t[:, mode, :P, :6]
            t0 = tt[:, :, 0].min(axis=1, keepdims=True)
            r = (tt - t0[:, :, None]).mean(axis=0)
            print(f"   {nm}: cycles from the team's start")
            print("      wave   start  element/recursion  boundary seen  hop posted  phase3 end  after barrier")
            for w in range(P):
                print(f"      {w:4d} " + " ".join(f"{v:12.0f}" for v in r[w]))
viol = np.zeros(1)
lib.rat_diag_read_n(ctx.h, viol.ctypes.data_as(C.POINTER(C.c_double)), 4095, 1)
print(f"team-barrier invariant violations counted by the diagnostic build (psweep.h): {int(viol[0])}")
