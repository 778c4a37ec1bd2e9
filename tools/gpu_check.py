"""First-contact GPU check: every operator of the C ABI against the CPU oracle (prints errors, never asserts)."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from ratilqr.jl_amd import ileqg as il
from oracle import oracle as orc

def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

def section(name, fn):
    try:
        t = time.time(); fn(); print(f"[ok ] {name} ({time.time()-t:.2f}s)")
    except Exception:
        print(f"[ERR] {name}"); traceback.print_exc()

def run(n, m, N, kappa, model="lq"):
    print(f"=== n={n} m={m} N={N} kappa={kappa} model={model}")
    if model == "lq":
        prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=3, kappa=kappa)
        u = 0.1 * np.random.default_rng(1).standard_normal((N, m))
    else:
        prob = rat.PowerLawRiskSensitiveProblem(n, N, 0.01 * np.eye(n)); x0 = np.zeros(n); u = 0.1 * np.ones((N, m))
    P = orc.Problem(prob)
    ctx = il.Context(prob, max_batch=64, spec_eps=4)
    st = {}
    def t_roll():
        _, xo = orc.simulate_open(P, x0, u); xg = ctx.rollout_open(x0, u); st["x"] = xo
        print("   rollout_open rel err", rel(xg, xo))
    section("rollout_open", t_roll)
    def t_fb():
        Lr = 0.1 * np.random.default_rng(2).standard_normal((N, m, n)); l2 = u + 0.01
        _, xo, uo = orc.simulate_feedback(P, st["x"], l2, Lr); xg, ug = ctx.rollout_feedback(st["x"], l2, Lr)
        print("   rollout_feedback rel err x", rel(xg, xo), "u", rel(ug, uo))
    section("rollout_feedback", t_fb)
    def t_cost():
        _, co = orc.integrate_cost(P, st["x"], u); cg = ctx.integrate_cost(st["x"], u); print("   integrate_cost", co, cg, abs(co-cg)/abs(co))
    section("integrate_cost", t_cost)
    def t_ap():
        _, apo = orc.approximate_model(P, u, st["x"]); a = apo.arrays(); st["apo"] = apo
        apg = ctx.approximate_model(u, st["x"]); st["apg"] = apg
        for k, kk in [("q","q_array"),("qv","q_vec_array"),("Q","Q_array"),("r","r_array"),("R","R_array"),("P","P_array"),("A","A_array"),("B","B_array"),("W","W_array")]:
            print(f"   approx {k}: {rel(getattr(apg, kk), a[k]):.2e}", end="")
        print()
    section("approximate_model", t_ap)
    for theta in (0.0, 2.0):
        def t_gain():
            _, Lo, dlo, dpo, muo, deo = orc.dp_gain(P, st["apo"], theta)
            stg, Lg, dlg, dpg, mug, deg = ctx.dp_gain_sweep(st["apg"], theta, 0.0, 2.0)
            st["Lo"] = Lo
            print(f"   gain theta={theta}: status {stg} L {rel(Lg, Lo):.2e} dl {rel(dlg, dlo):.2e} s {rel(dpg.s_array, dpo['s']):.2e} "
                  f"S {rel(dpg.S_array, dpo['S']):.2e} sv {rel(dpg.s_vec_array, dpo['sv']):.2e} g {rel(dpg.g_array, dpo['g']):.2e} "
                  f"G {rel(dpg.G_array, dpo['G']):.2e} H {rel(dpg.H_array, dpo['H']):.2e} mu {mug} {muo}")
            if rel(dpg.s_array, dpo['s']) > 1e-8:
                print("   s gpu", dpg.s_array[-4:], "\n   s orc", dpo['s'][-4:])
                print("   S gpu[N-1]\n", dpg.S_array[N-1][:4,:4], "\n   S orc[N-1]\n", dpo['S'][N-1][:4,:4])
                print("   H gpu[N-1]\n", dpg.H_array[N-1], "\n   H orc\n", dpo['H'][N-1])
                print("   G gpu[N-1]\n", dpg.G_array[N-1][:, :4], "\n   G orc\n", dpo['G'][N-1][:, :4])
        section(f"dp_gain_sweep theta={theta}", t_gain)
        def t_eval():
            Lt = st["Lo"] * 0.9
            _, dpo = orc.dp_eval(P, st["apo"], Lt, None, theta, 1e-6)
            stg, dpg = ctx.dp_policy_eval(st["apg"], Lt, None, theta, 1e-6)
            print(f"   eval theta={theta}: status {stg} s {rel(dpg.s_array, dpo['s']):.2e} S {rel(dpg.S_array, dpo['S']):.2e}")
        section(f"dp_policy_eval theta={theta}", t_eval)
    def t_solve():
        for theta in (0.0, 1.0):
            so = orc.ILEQGSolver(P); rc = so.solve(x0, u, theta)
            r = ctx.solve(x0, u, theta)
            print(f"   solve theta={theta}: oracle rc {rc} val {so.s.value_current!r} it {so.s.iter_current} | gpu st {r['status']} val {r['value']!r} it {r['iters']} "
                  f"x {rel(r['x'], so.x_array):.2e} l {rel(r['l'], so.l_array):.2e} L {rel(r['L'], so.L_array):.2e} hist {r['eps_history'][:3].tolist()} / {so.eps_history[:3].tolist()}")
    section("solve", t_solve)
    def t_batch():
        th = np.concatenate([[0.0], np.linspace(0.01, 14.0, 62), [50.0]]) if model == "lq" else np.array([0.0, 0.1, 0.3, 0.43, 0.5])
        vo, so, io, lo = orc.compute_value_batch(P, x0, u, th, nthreads=8)
        t = time.time(); vg, sg, ig, lg = ctx.solve_batch(x0, u, th); dt = time.time() - t
        fin = np.isfinite(vo)
        print(f"   batch B={th.size}: status match {np.array_equal(so, sg)} iters match {np.array_equal(io, ig)} ls match {np.array_equal(lo, lg)} "
              f"finite match {np.array_equal(fin, np.isfinite(vg))} value rel {rel(vg[fin], vo[fin]):.2e}  ({dt*1e3:.1f} ms)")
        if not np.array_equal(so, sg): print("   status oracle", so, "\n   status gpu   ", sg)
    section("solve_batch", t_batch)

if __name__ == "__main__":
    run(12, 4, 50, 0.0)
    run(12, 4, 50, 0.02)
    run(4, 2, 20, 0.0)
    run(2, 2, 10, 0.0, model="pl")
