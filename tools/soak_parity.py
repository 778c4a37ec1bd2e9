"""Randomised parity soak of the batched solve against the CPU oracle: SOAK_N random LQ problems (shapes 1..12 x 1..4 x 1..60, random
time-varying tables, cubic drift, random solver options and speculation widths, every third problem zoomed on its feasibility
boundary).  Round 1: 3000 problems, 2 mismatches, both on trajectories that overflow to 1e57+ and run into iter_max (chaotic
line-search paths / an LU singularity in 1e120-scale arithmetic): no mismatch on a well-posed problem.
  SOAK_N=3000 python tools/soak_parity.py   (on an MI355X; ~40 s)
  SOAK_WIDE=16: n 13..16 with m 1..4 (the register sweeps and rollouts of wide16.h).
  SOAK_WIDE=1: shapes beyond the tile instead -- n 13..32 with m 1..32, or n 1..12 with m 5..32; N 1..30 (the general-size kernels of wide.hip)."""
import sys, os, time
os.environ.setdefault("RATILQR_SPEC_FORCE", "1")     # handles of width E > 1 run the speculative kernels here (spec_eps is otherwise an upper bound)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ratilqr.jl_amd as rat
from oracle import oracle as orc
bad = 0
DUMP = os.environ.get('SOAK_DUMP')              # SOAK_DUMP=file.npz: every problem's device results, for bit-wise comparison between two builds
dump = {}                                       # (tools/novf_diff.sh)
t0 = time.time()
NS = int(os.environ.get('SOAK_N', '240'))
S0 = int(os.environ.get('SOAK_SEED0', '0'))     # first problem seed: SOAK_SEED0=3000 SOAK_N=3000 runs problems 3000..5999
SEEDS = [int(v) for v in os.environ['SOAK_SEEDS'].split(',')] if os.environ.get('SOAK_SEEDS') else range(S0, S0 + NS)   # SOAK_SEEDS=4228,6308: those problems only
for seed in SEEDS:
    rng = np.random.default_rng(5000 + seed)
    n, m, N = int(rng.integers(1, 13)), int(rng.integers(1, 5)), int(rng.integers(1, 61))
    if os.environ.get('SOAK_WIDE') == '1':
        n, m, N = int(rng.integers(13, 33)), int(rng.integers(1, 13)), int(rng.integers(1, 31))
        if seed % 3 == 1:                      # few states, many controls (only m exceeds the tile); up to the largest m
            n, m = int(rng.integers(1, 13)), int(rng.integers(5, 33))
        elif seed % 3 == 2:
            m = int(rng.integers(1, 33))
    if os.environ.get('SOAK_WIDE') == '16':   # the register sweep of wide16.h: n 13..16, m 1..4, N 1..40
        n, m, N = int(rng.integers(13, 17)), int(rng.integers(1, 5)), int(rng.integers(1, 41))
    tv = bool(rng.integers(0, 2))
    A = (0.7 + 0.3 * rng.random()) * np.linalg.qr(rng.standard_normal((n, n)))[0]
    B = rng.standard_normal((n, m)) / np.sqrt(n)
    def spd(k, scale):
        G = rng.standard_normal((k, k))
        return scale * (np.eye(k) + 0.2 * G @ G.T / k)
    if tv:
        Q = np.stack([spd(n, 0.5 + rng.random()) for _ in range(N)])
        R = np.stack([spd(m, 0.1 + 0.3 * rng.random()) for _ in range(N)])
        Pm = 0.03 * rng.standard_normal((N, m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal((N, n)), 0.1 * rng.standard_normal((N, m)), rng.standard_normal(N)
        W = np.stack([spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)])
    else:
        Q, R, Pm = spd(n, 1.0), spd(m, 0.2), 0.03 * rng.standard_normal((m, n))
        qv, rv, q0 = 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(m), float(rng.standard_normal())
        W = spd(n, 1e-3)
        if seed % 4 == 1:                      # diagonal W (the sweeps' inv(W)-folded arithmetic, ProblemDev.W_diag), entries over a decade
            W = np.diag(1e-3 * 10.0 ** rng.uniform(-0.5, 0.5, n))
    kappa = float(rng.choice([0.0, 0.02, -0.02, 0.04]))
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Q, R=R, P=Pm, qv=qv, rv=rv, q0=q0, N=N, W=W, Qf=spd(n, 1.0),
                                      qvf=0.2 * rng.standard_normal(n), q0f=float(rng.standard_normal()), kappa=kappa)
    x0, u = rng.uniform(0.3, 1.0) * rng.standard_normal(n), 0.1 * rng.standard_normal((N, m))
    theta = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-2, 2.5, 11))])
    if seed % 3 == 0:      # zoom on the feasibility boundary of this problem
        grid = 10.0 ** np.linspace(-2, 3, 41)
        _, sgrid, _, _ = orc.compute_value_batch(orc.Problem(prob), x0, u, grid, nthreads=16)
        badi = np.nonzero(sgrid != 0)[0]
        if badi.size and badi[0] > 0:
            lo_, hi_ = grid[badi[0] - 1], grid[badi[0]]
            theta = np.concatenate([[0.0], np.linspace(lo_, hi_, 23)])
    E = int(rng.choice([1, 1, 2, 3, 4, 8, 11]))       # 1, 2, 4, 8: solve_block_kernel (default path for small batches); 3, 11: round-based path
    if os.environ.get('SOAK_E'):                      # SOAK_E=1: every problem on the E = 1 default path (small batches: the time-parallel sweeps of
        E = int(os.environ['SOAK_E'])                 # solve_block_psw_kernel)
    kw = {}
    if rng.integers(0, 2):
        kw = dict(lam=float(rng.uniform(0.3, 0.7)), iter_max=int(rng.integers(3, 40)), adaptive_eps_init=int(rng.integers(0, 2)))
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=16, **kw)
    gkw = dict(kw)
    if "adaptive_eps_init" in gkw: gkw["adaptive_eps_init"] = bool(gkw["adaptive_eps_init"])
    ctx = rat.Context(prob, rat.ileqg.make_opts(**gkw), max_batch=theta.size, spec_eps=E)
    if not os.environ.get('SOAK_WIDE') and seed % 2 == 1 and not os.environ.get('SOAK_E'):
        ctx.set_path("rounds")                 # E > 1: candidates without tile records (fly sweeps, all candidates of a sample in one rollout wave)
    vg, sg, ig, lg = ctx.solve_batch(x0, u, theta)
    if DUMP:
        dump[f"v{seed}"], dump[f"s{seed}"], dump[f"i{seed}"], dump[f"l{seed}"] = vg, sg, ig, lg
    fin = np.isfinite(vo)
    ok = np.array_equal(sg, so) and np.array_equal(ig, io) and np.array_equal(lg, lo) and np.array_equal(fin, np.isfinite(vg)) \
        and (not fin.any() or np.all(np.abs(vg[fin] - vo[fin]) <= 1e-9 * np.abs(vo[fin])))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, (n, m, N), tv, kappa, E, kw, so.tolist(), sg.tolist(), io.tolist(), ig.tolist(), lo.tolist(), lg.tolist())
        if os.environ.get('SOAK_SEEDS'):     # diagnosis: where the two differ, with the values
            for i in np.nonzero((so != sg) | (io != ig) | (lo != lg) | ~((vo == vg) | (np.abs(vo - vg) <= 1e-9 * np.abs(vo)) | (~np.isfinite(vo) & ~np.isfinite(vg))))[0]:
                print(f"   sample {i} theta {theta[i]:.6g}: oracle value {vo[i]:.15g} status {so[i]} iters {io[i]} ls {lo[i]} | device value {vg[i]:.15g} status {sg[i]} iters {ig[i]} ls {lg[i]}")
if DUMP:
    np.savez(DUMP, **dump)
print("soak done:", NS, "problems,", bad, "mismatches,", round(time.time() - t0, 1), "s")
