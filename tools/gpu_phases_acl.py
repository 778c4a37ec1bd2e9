"""Role timeline of the deviation-form rollout inside solve_block_kernel (diagp build): per hardware wave, cycles from the staging barrier
to the end of its recursion / production part and to the end of its linearising part, and the steps it linearised.  python tools/gpu_phases_acl.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RATILQR_SO"] = os.path.join(ROOT, "ratilqr.jl_amd", "csrc", "libratilqr_hip_diagp.so")
os.environ["RATILQR_BLOCK"] = "1"
sys.path.insert(0, ROOT)
import numpy as np
import ratilqr.jl_amd as rat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prob, x0, u = rat.synthetic_lq_problem()
ctx = rat.Context(prob, max_batch=B, spec_eps=1)
lib = rat.native.lib()
lib.rat_diag_read_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64]
for _ in range(3):
    ctx.solve_batch(x0, u, np.full(B, 1.0))
out = np.zeros(512)
lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 2048, 512)
t = out.reshape(8, 4, 16)
print(f"B = {B}: last closed-loop rollout, hardware waves 0..3 (mean over 8 samples), cycles from the wave's own start stamp")
for w in range(4):
    s0, s1, s2, n = t[:, w, 0], t[:, w, 1], t[:, w, 2], t[:, w, 3]
    if not s0.any():
        continue
    print(f"   wave {w}: first part (recursion / production / -) {np.mean(s1 - s0):8.0f}   linearising {np.mean(s2 - s1):8.0f}   steps linearised {np.mean(n):5.1f}   total {np.mean(s2 - s0):8.0f}")
seg = t[:, 0, 4:8].mean(0)
print(f"   recursion wave, cycles per rollout by segment: post + cubic + accumulator start {seg[0]:.0f} | poll + operand requests {seg[1]:.0f} | 3 MFMAs + dx {seg[2]:.0f} | loop overhead between steps {seg[3]:.0f}")
