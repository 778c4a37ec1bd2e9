"""Time-parallel sweep (csrc/psweep.h) against the sequential sweep kernels on the batched sweep operators, one GPU.

For B trajectories of the headline problem (N = 50, n = 12, m = 4; thetas spread over the feasible range) runs rat_dp_gain_sweep_batch /
rat_dp_policy_eval_batch with the switch psweep = 0 (sweep_kernel: one wavefront per trajectory) and psweep = P (psweep_kernel: P
wavefronts per trajectory), compares the results and prints kernel times from the library's own HIP events.
    python tools/psweep_time.py [B ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import _native as nv
from ratilqr.jl_amd import ileqg as il
from ratilqr.jl_amd.generic import GenericContext
from oracle import oracle as orc


def approx_of(prob, x0, u):
    P = orc.Problem(prob)
    _, x = orc.simulate_open(P, x0, u)
    _, ap = orc.approximate_model(P, u, x)
    a = ap.arrays()
    return P, ap, il.ApproximationResult(q_array=a["q"], q_vec_array=a["qv"], Q_array=a["Q"], r_array=a["r"], R_array=a["R"], P_array=a["P"],
                                         A_array=a["A"], B_array=a["B"], W_array=a["W"])


class Harness:
    def __init__(self, prob, B):
        self.prob, self.B = prob, B
        self.ctx = rat.Context(prob, max_batch=B)
        self.n, self.m, self.N = prob.n, prob.m, prob.N

    def gain(self, ap, theta, mu=None, delta=None, P=0, **sw):
        B, n, m, N = self.B, self.n, self.m, self.N
        self.ctx.debug_set("psweep", P)
        for k, v in sw.items():
            self.ctx.debug_set(k, v)
        bufs = GenericContext._stack([ap] * B)
        th = nv.f64(theta).copy()
        mu_c = np.zeros(B) if mu is None else nv.f64(mu).copy()
        de_c = 2.0 * np.ones(B) if delta is None else nv.f64(delta).copy()
        Lb, dl, st = np.zeros(B * m * n * N), np.zeros((B, N, m)), np.zeros(B, np.int32)
        self.ctx.profile(True); self.ctx.profile_reset()
        nv.check(nv.lib().rat_dp_gain_sweep_batch(self.ctx.h, C.c_int64(B), *[nv.P(b) for b in bufs], nv.P(th), nv.P(mu_c), nv.P(de_c), nv.P(Lb),
                                                  nv.P(dl), nv.PI(st)))
        ms = self.ctx.profile_get()["sweep_gain"]["ms"]
        self.ctx.profile(False)
        L = np.stack([nv.from_cm3(Lb[b * m * n * N:(b + 1) * m * n * N], N, m, n) for b in range(B)])
        return dict(st=st, L=L, dl=dl, mu=mu_c, delta=de_c, ms=ms)

    def evalp(self, ap, Ls, theta, mu, P=0, **sw):
        B = self.B
        self.ctx.debug_set("psweep", P)
        for k, v in sw.items():
            self.ctx.debug_set(k, v)
        bufs = GenericContext._stack([ap] * B)
        Lc = np.concatenate([nv.cm3(L) for L in Ls])
        val, st = np.zeros(B), np.zeros(B, np.int32)
        self.ctx.profile(True); self.ctx.profile_reset()
        nv.check(nv.lib().rat_dp_policy_eval_batch(self.ctx.h, C.c_int64(B), *[nv.P(b) for b in bufs], nv.P(Lc), nv.P(nv.f64(theta)), nv.P(nv.f64(mu)),
                                                   nv.P(val), nv.PI(st)))
        ms = self.ctx.profile_get()["sweep_eval"]["ms"]
        self.ctx.profile(False)
        return dict(val=val, st=st, ms=ms)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [128, 512]
    prob, x0, _ = rat.synthetic_lq_problem()
    u = 0.1 * np.random.default_rng(1).standard_normal((prob.N, prob.m))
    Pp, ap_o, ap = approx_of(prob, x0, u)
    out = {}
    for B in Bs:
        hs = Harness(prob, B)
        theta = np.linspace(0.05, 12.0, B)
        ref = hs.gain(ap, theta)
        ref = hs.gain(ap, theta)                         # (second call: warm)
        Ls = 0.9 * ref["L"]
        mu = 1e-6 * np.ones(B)
        refe = hs.evalp(ap, Ls, theta, mu)
        refe = hs.evalp(ap, Ls, theta, mu)
        # oracle on a few samples
        for b in (0, B // 2, B - 1):
            _, Lo, dlo, dpo, _, _ = orc.dp_gain(Pp, ap_o, float(theta[b]))
            _, dpe = orc.dp_eval(Pp, ap_o, Ls[b], None, float(theta[b]), 1e-6)
            assert rel(ref["L"][b], Lo) < 1e-10 and abs(refe["val"][b] - dpe["s"][0]) < 1e-10 * abs(dpe["s"][0])
        row = {"seq": {"gain_ms": ref["ms"], "eval_ms": refe["ms"]}}
        print(f"B={B}: sequential gain {ref['ms']:.4f} ms, eval {refe['ms']:.4f} ms", flush=True)
        grid = [(int(h), int(c)) for h, c in (g.split(":") for g in os.environ.get("PSW_GRID", "130:125").split(","))]
        for P in [int(p) for p in os.environ.get("PSW_P", "2,3,4,5,6,8").split(",")]:
            for hop, comp in grid:
                g = hs.gain(ap, theta, P=P, psw_hop=hop, psw_comp=comp)
                g = hs.gain(ap, theta, P=P, psw_hop=hop, psw_comp=comp)
                e = hs.evalp(ap, Ls, theta, mu, P=P)
                e = hs.evalp(ap, Ls, theta, mu, P=P)
                okg = np.array_equal(g["st"], ref["st"]) and np.array_equal(g["mu"], ref["mu"])
                eL, edl, ev = rel(g["L"], ref["L"]), rel(g["dl"], ref["dl"]), rel(e["val"], refe["val"])
                oke = np.array_equal(e["st"], refe["st"])
                row[f"P{P}_h{hop}_c{comp}"] = {"gain_ms": g["ms"], "eval_ms": e["ms"], "err_L": eL, "err_dl": edl, "err_val": ev, "status_equal": bool(okg and oke)}
                print(f"   P={P} (hop {hop}, comp {comp}): gain {g['ms']:.4f} ms  eval {e['ms']:.4f} ms | err L {eL:.1e} dl {edl:.1e} val {ev:.1e} status_equal {okg and oke}",
                      flush=True)
        out[str(B)] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
