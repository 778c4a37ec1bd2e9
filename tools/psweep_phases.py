"""Phase timeline of the time-parallel sweep (psweep_kernel): cycles at which every wave of a trajectory's workgroup finishes its element
(phase 1; the last wave: its own recursion), sees the boundary value, posts its hop and ends its ordinary pass (phase 3).
Build: make -C ratilqr.jl_amd/csrc diagp.   python tools/psweep_phases.py [B] [P ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diagp.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import ratilqr.jl_amd as rat
from psweep_time import Harness, approx_of

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Ps = [int(a) for a in sys.argv[2:]] or [4, 5, 8]
prob, x0, _ = rat.synthetic_lq_problem()
u = 0.1 * np.random.default_rng(1).standard_normal((prob.N, prob.m))
Pp, ap_o, ap = approx_of(prob, x0, u)
hs = Harness(prob, B)
lib = rat.native.lib()
lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
theta = np.linspace(0.05, 12.0, B)
ref = hs.gain(ap, theta)
Ls, mu = 0.9 * ref["L"], 1e-6 * np.ones(B)
hop, comp = int(os.environ.get("PSW_HOP", 130)), int(os.environ.get("PSW_COMP", 125))
for P in Ps:
    for kind in ("gain", "eval"):
        for _ in range(3):
            r = hs.gain(ap, theta, P=P, psw_hop=hop, psw_comp=comp) if kind == "gain" else hs.evalp(ap, Ls, theta, mu, P=P, psw_hop=hop, psw_comp=comp)
        out = np.zeros(4096)
        lib.rat_diag_read_n(hs.ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 4096, 4096)
        t = out.reshape(8, 8, 4, 16)[:, 0 if kind == "gain" else 1, :min(P, 4), :6]          # (mode 0: gain sweep, mode 1: evaluation; waves 0..3)
        t0 = t[:, :, 0].min(axis=1, keepdims=True)
        rel = (t - t0[:, :, None]).mean(axis=0)
        print(f"{kind} P={P} (model hop {hop} comp {comp}): kernel {r['ms'] * 1e3:.1f} us; cycles from the first wave's start, mean of 8 trajectories")
        print("   wave   start  phase1/recursion  boundary seen  hop posted  phase3 end  after barrier")
        for w in range(P):
            print(f"   {w:4d} " + " ".join(f"{v:12.0f}" for v in rel[w]))
