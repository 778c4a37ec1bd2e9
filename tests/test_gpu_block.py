"""solve_block_kernel -- the whole solve! of a theta-sample by ONE WORKGROUP (a wavefront per speculative line-search candidate plus a
gain-sweep wavefront) -- against the other execution paths (bit for bit: it calls the same device functions) and against the oracle
(counts identical, values 1e-9), on workloads whose line search really backtracks."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from oracle import oracle as orc
from test_gpu_parity import check_batch, stress_problem

pytestmark = pytest.mark.gpu


def _workloads():
    back, bx0, bu = rat.synthetic_lq_problem(kappa=0.06)                      # 5 iterations, 6..13 line-search evaluations
    th_b = np.array([0.0, 0.5, 2.0, 5.0, 6.5, 8.0, 30.0])
    lq, lx0, lu = rat.synthetic_lq_problem()
    th_l = np.concatenate([[0.0], np.linspace(0.01, 14.0, 30), [50.0]])
    stress = [stress_problem(i, kappa=0.03) for i in range(4)]
    th_s = np.array([0.0, 0.3, 1.0, 4.0])
    pl = rat.PowerLawRiskSensitiveProblem(2, 10, 0.01 * np.eye(2), a=1.3, b=1.5, p=2.5, hconst=1.0)
    return back, bx0, bu, th_b, lq, lx0, lu, th_l, stress, th_s, pl


def _run_all(E):
    back, bx0, bu, th_b, lq, lx0, lu, th_l, stress, th_s, pl = _workloads()
    out = []
    out += rat.Context(back, max_batch=th_b.size, spec_eps=E).solve_batch(bx0, bu, th_b)
    out += rat.Context(lq, max_batch=32, spec_eps=E).solve_batch(lx0, lu, th_l)
    for sp, sx, su in stress:
        out += rat.Context(sp, rat.ileqg.make_opts(iter_max=8), max_batch=4, spec_eps=E).solve_batch(sx, su, th_s)
    out += rat.Context(pl, max_batch=3, spec_eps=E).solve_batch(np.zeros(2), 0.1 * np.ones((10, 2)), np.array([0.0, 0.5, 2.0]))
    r = rat.Context(back, spec_eps=E).solve(bx0, bu, 5.0)
    out += [r["L"], r["x"], r["l"], np.array([r["value"], r["status"], r["iters"]]), np.asarray(r["eps_history"], dtype=float)]
    return out


@pytest.mark.parametrize("E", [1, 2, 4, 8])
def test_block_kernel_is_bit_identical_to_the_other_paths(E, monkeypatch):
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    monkeypatch.setenv("RATILQR_BLOCK", "1")
    block = _run_all(E)
    monkeypatch.setenv("RATILQR_BLOCK", "0")
    other = _run_all(E)                                        # E = 1: solve_fused_kernel; E > 1: round-based path
    monkeypatch.setenv("RATILQR_FUSED", "0")
    rounds = _run_all(E)
    assert len(block) == len(other) == len(rounds)
    for a, b, c in zip(block, other, rounds):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        assert np.array_equal(np.asarray(a), np.asarray(c), equal_nan=True)


@pytest.mark.parametrize("E", [1, 2, 4, 8])
def test_block_kernel_against_the_oracle_on_a_backtracking_workload(E, monkeypatch):
    monkeypatch.setenv("RATILQR_BLOCK", "1")
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.06)
    P = orc.Problem(prob)
    theta = np.array([0.0, 0.5, 2.0, 3.5, 5.0, 6.0, 7.5, 8.0, 20.0])
    ctx = rat.Context(prob, max_batch=theta.size, spec_eps=E)
    assert ctx.profile_get()["solve_block"]["launches"] == 0
    ctx.profile(True)
    vg, sg, ig, lg = check_batch(ctx, P, x0, u, theta)
    assert ctx.profile_get()["solve_block"]["launches"] == 1 and ctx.profile_get()["solve_fused"]["launches"] == 0
    ctx.profile(False)
    assert lg.max() >= 10 and (lg > ig).any()                  # the line search really backtracks on this problem
    # eps history of a single solve: the sequence of accepted / rejected step sizes is the sequential one
    r = ctx.solve(x0, u, 5.0)
    s = orc.ILEQGSolver(P)
    assert s.solve(x0, u, 5.0) == r["status"] == 0
    ho = s.eps_history
    assert r["eps_history"].shape == ho.shape and np.array_equal(r["eps_history"][:, 0], ho[:, 0])
    assert np.abs(r["eps_history"][:, 1] - ho[:, 1]).max() <= 1e-9 * max(1.0, np.abs(ho[:, 1]).max())
    assert np.abs(r["L"] - s.L_array).max() < 1e-9 and np.abs(r["x"] - s.x_array).max() < 1e-9


@pytest.mark.parametrize("lam,eps_min", [(0.9, 1e-6), (0.8, 0.3), (0.5, 1e-6)])
def test_eight_candidates_with_a_lazily_evaluated_last_one(lam, eps_min, monkeypatch):
    """E = 8: the workgroup's eighth wave runs the gain sweeps beside the evaluations of candidates 0..6 and evaluates candidate 7 only when
    the accept rule rejects all of them.  With lambda = 0.9 a backtracking line search rejects more than eight step sizes in a row (rounds
    in which the last candidate is needed, and whole rounds without an accepted candidate); eps_min = 0.3 forces an accept inside a round.
    Same counts and bits as the round-based path, counts equal to the oracle's and values to 1e-9."""
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.06)
    theta = np.array([0.0, 0.5, 2.0, 3.5, 5.0, 6.0, 7.5, 8.0, 20.0])
    opts = rat.ileqg.make_opts(lam=lam, eps_min=eps_min, iter_max=12)
    monkeypatch.setenv("RATILQR_BLOCK", "1")
    ctx = rat.Context(prob, opts, max_batch=theta.size, spec_eps=8)
    ctx.profile(True)
    blk = ctx.solve_batch(x0, u, theta)
    assert ctx.profile_get()["solve_block"]["launches"] == 1
    monkeypatch.setenv("RATILQR_BLOCK", "0")
    rnd = rat.Context(prob, opts, max_batch=theta.size, spec_eps=8).solve_batch(x0, u, theta)
    for a, b in zip(blk, rnd):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    monkeypatch.setenv("RATILQR_BLOCK", "1")
    P = orc.Problem(prob)
    kw = dict(lam=lam, eps_min=eps_min, iter_max=12)
    vg, sg, ig, lg = check_batch(ctx, P, x0, u, theta, **kw)  # statuses, iterations, line-search evaluations equal; values to 1e-9
    for a, b in zip((vg, sg, ig, lg), blk):
        assert np.array_equal(a, b, equal_nan=True)
    deep = 0
    for th in (2.0, 6.0, 8.0):
        s = orc.ILEQGSolver(P, **kw)
        r = ctx.solve(x0, u, th)
        assert s.solve(x0, u, th) == r["status"]
        h = s.eps_history
        assert r["eps_history"].shape == h.shape and np.array_equal(r["eps_history"][:, 0], h[:, 0])
        deep += int(np.sum(h[:, 0] <= lam ** 7 * (1 + 1e-12)))
    if lam == 0.9:
        assert deep > 0                                        # step sizes at or beyond the eighth candidate of a round were evaluated


def test_block_geometries_are_bit_identical(monkeypatch):
    """E = 1 geometries of the block kernel: four-wave workgroups with ticketed SIMD pairs and linearise helper waves (one sample per CU),
    without the helpers, plain two-wave workgroups -- and rollouts split over the waves or not (N > 52: unsplit)."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths; the time-parallel
                                                               # sweeps of small batches agree to rounding (tests/test_gpu_psweep.py)
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.04)
    theta = np.concatenate([[0.0], np.linspace(0.05, 9.0, 60), [40.0]])
    long_p, lx0, lu = rat.synthetic_lq_problem(N=60, seed=3, kappa=0.02)
    th_l = np.array([0.0, 0.5, 2.0])

    def run():
        out = list(rat.Context(prob, max_batch=theta.size).solve_batch(x0, u, theta))
        out += list(rat.Context(long_p, max_batch=3).solve_batch(lx0, lu, th_l))
        r = rat.Context(prob).solve(x0, u, 3.0)
        return out + [r["x"], r["l"], r["L"], np.array([r["value"]]), np.asarray(r["eps_history"], dtype=float)]

    monkeypatch.setenv("RATILQR_BLOCK", "1")
    ref = run()                                                # padded workgroups + helpers
    for env in ({"RATILQR_BLOCK_HELPERS": "0"}, {"RATILQR_BLOCK_SHAPE": "0"}, {"RATILQR_BLOCK": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k) if k != "RATILQR_BLOCK" else monkeypatch.setenv("RATILQR_BLOCK", "1")
        assert all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(ref, got)), env
    P = orc.Problem(prob)
    check_batch(rat.Context(prob, max_batch=theta.size), P, x0, u, theta)


def test_block_kernel_default_policy_and_several_generations(monkeypatch):
    """E = 1: batches up to 512 samples run on the block kernel (two SIMDs per sample), larger ones on the fused kernel; forced, a batch
    larger than the chip runs its workgroups in several generations with unchanged results."""
    monkeypatch.setenv("RATILQR_BLOCK_PSW", "0")               # bit-identity is a property of the sequential-sweep paths (the two-wave kernel runs the
                                                               # evaluation that ends a solve time-parallel under block_psw: equal to rounding)
    prob, x0, u = rat.synthetic_lq_problem()
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(9).standard_normal(2500))
    ctx = rat.Context(prob, max_batch=2500)
    ctx.profile(True)
    small = ctx.solve_batch(x0, u, theta[:512])
    p1 = ctx.profile_get()
    assert p1["solve_block"]["launches"] == 1 and p1["solve_fused"]["launches"] == 0
    big = ctx.solve_batch(x0, u, theta)
    p2 = ctx.profile_get()
    assert p2["solve_block"]["launches"] == 1 and p2["solve_fused"]["launches"] == 1
    monkeypatch.setenv("RATILQR_BLOCK", "1")
    forced = rat.Context(prob, max_batch=2500).solve_batch(x0, u, theta)
    for a, b in zip(big, forced):
        assert np.array_equal(a, b)
    for a, b in zip(small, big):
        assert np.array_equal(a, b[:512])


def test_padded_workgroups_with_a_co_resident_kernel():
    """The padded two-wave geometry (four waves per workgroup, roles read off the hardware SIMD ids) assumes the four waves of a workgroup
    sit on four distinct SIMDs; with another kernel resident on the CU (a second handle on the device, RCCL, a framework kernel) the
    dispatcher does not guarantee that, and the kernel then falls back to roles by wave index.  Two host threads drive two handles (one
    thread per handle, as include/ratilqr.h asks) with batches of 300 and 256 samples at the same time -- their launches overlap on the
    device -- and every output must equal the one of the handle run alone."""
    import threading
    prob, x0, u = rat.synthetic_lq_problem()
    back, bx0, bu = rat.synthetic_lq_problem(kappa=0.06)
    rng = np.random.default_rng(8)
    th1 = np.abs(1.0 + 2.0 * rng.standard_normal(300)); th1[::37] = 40.0
    th2 = np.concatenate([[1e-3, 0.5, 2.0, 5.0, 6.5, 8.0, 30.0], 6.0 * rng.random(249)])
    c1 = rat.Context(prob, max_batch=th1.size)
    c2 = rat.Context(back, max_batch=th2.size)
    assert c1.get_path(th1.size) == "block" and c2.get_path(th2.size) == "block"
    ref1, ref2 = c1.solve_batch(x0, u, th1), c2.solve_batch(bx0, bu, th2)
    assert np.isfinite(ref1[0]).sum() > 250 and np.isfinite(ref2[0]).sum() > 200 and ref2[2].max() >= 4
    bad = []

    def drive(ctx, a, b, th, ref, tag):
        for rep in range(12):
            got = ctx.solve_batch(a, b, th)
            for q, (x, y) in enumerate(zip(got, ref)):
                if not np.array_equal(x, y):
                    bad.append((tag, rep, q))

    t1 = threading.Thread(target=drive, args=(c1, x0, u, th1, ref1, "lq"))
    t2 = threading.Thread(target=drive, args=(c2, bx0, bu, th2, ref2, "backtracking"))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert not bad, bad[:8]


@pytest.mark.parametrize("B,kappa", [(96, 0.0), (300, 0.0), (64, 0.05)])
def test_deviation_form_rollouts_agree_to_rounding(B, kappa):
    """Switch block_acl = 1 (opt-in): the split geometry's closed-loop rollouts as dx_{t+1} = (A + B L_t) dx_t + eps B dl_t + drift terms
    on the recursion wave (3 MFMAs per step), controls and cost rows by the linearising waves from a step pool.  Its own rounding order:
    values agree with the default (bit-identical) paths to ~1e-15, every count is equal, parity against the oracle is unchanged."""
    prob, x0, u = rat.synthetic_lq_problem(kappa=kappa)
    theta = np.concatenate([[0.0], np.abs(1.0 + 2.0 * np.random.default_rng(B).standard_normal(B - 2)) + 0.01, [40.0]])
    res = {}
    for acl in (0, 1):
        ctx = rat.Context(prob, max_batch=B)
        ctx.debug_set("block_acl", acl)
        assert ctx.get_path(B) == "block" and ctx.debug_get("block_acl") == acl
        res[acl] = ctx.solve_batch(x0, u, theta)
    (v0, s0, i0, l0), (v1, s1, i1, l1) = res[0], res[1]
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1)
    fin = np.isfinite(v0)
    assert np.array_equal(fin, np.isfinite(v1)) and np.abs(v1[fin] - v0[fin]).max() <= 1e-12 * np.abs(v0[fin]).max()
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta[:24], nthreads=8)
    f = np.isfinite(vo)
    assert np.array_equal(s1[:24], so) and np.array_equal(i1[:24], io) and np.array_equal(l1[:24], lo)
    assert np.abs(v1[:24][f] - vo[f]).max() <= 1e-9 * np.abs(vo[f]).max()
