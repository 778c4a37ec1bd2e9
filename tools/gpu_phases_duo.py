"""Timeline of solve_block_psw_kernel with two workgroups per sample (switch psw_duo): role A's waves 0 / 1 at every barrier and role B's wave 0
around its check-in, sweeps and waits -- s_memrealtime stamps (100 MHz, one clock for both compute units), first 8 samples of a batch
(build: make -C ratilqr.jl_amd/csrc diagp).  [PRL=0] python tools/gpu_phases_duo.py [B]"""
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
for duo in (0, 1):
    ctx = rat.Context(prob, max_batch=B, spec_eps=1)
    ctx.debug_set("psw_duo", duo)
    ctx.debug_set("psw_prl", int(os.environ.get("PRL", "1")))      # PRL=0: the one-wave closed-loop rollout
    for _ in range(3):
        ctx.solve_batch(x0, u, np.full(B, 1.0))
    out = np.zeros(1280)
    lib.rat_diag_read_n(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 1024, 1280)
    t = out.reshape(2, 8, 2, 40)
    print(f"psw_duo = {duo}, B = {B}; pairs formed: {ctx.debug_get('psw_duo_count')}")
    for smp in (0, 5):
        if smp >= B:
            continue
        a = t[0, smp, 0]; bb = t[1, smp, 0]
        t0 = a[0]
        na = int(np.max(np.nonzero(a)[0])) + 1
        print(f"  sample {smp}: role A wave 0 marks (us from its first): " + " ".join(f"{(v - t0) / 100:.1f}" for v in a[:na]))
        a1 = t[0, smp, 1]
        if os.environ.get("WAVE1") and a1.any():
            n1 = int(np.max(np.nonzero(a1)[0])) + 1
            print(f"  sample {smp}: role A wave 1 marks (us, same origin):  " + " ".join(f"{(v - t0) / 100:.1f}" for v in a1[:n1]))
        if duo and bb.any():
            nb = int(np.max(np.nonzero(bb)[0])) + 1
            print(f"  sample {smp}: role B wave 0 marks (us, same origin):  " + " ".join(f"{(v - t0) / 100:.1f}" for v in bb[:nb]))
viol = np.zeros(1)
lib.rat_diag_read_n(ctx.h, viol.ctypes.data_as(C.POINTER(C.c_double)), 4095, 1)
print(f"team-barrier invariant violations counted by the diagnostic build (psweep.h): {int(viol[0])}")
if int(os.environ.get("PRL", "1")):
    # rollprl_body's stamps of the LAST rollout of the solve (100 MHz): entry, element built, box taken, hop posted, segment done, end -- per wave
    pr = np.zeros(8 * 64)
    lib.rat_diag_read_n(ctx.h, pr.ctypes.data_as(C.POINTER(C.c_double)), 3072, 8 * 64)
    pr = pr.reshape(8, 4, 16)
    c = ctx.debug_get("prl_cuts")
    print(f"  rollprl cuts: 0 {c & 0xffff} {(c >> 16) & 0xffff} {(c >> 32) & 0xffff} {prob.N}")
    for smp in (0, 5):
        if smp >= B:
            continue
        o = pr[smp, 0, 0]
        for w in range(4):
            print(f"  sample {smp} rollprl wave {w} (us from wave 0's entry): " + " ".join(f"{(v - o) / 100:.2f}" for v in pr[smp, w, :6]))
