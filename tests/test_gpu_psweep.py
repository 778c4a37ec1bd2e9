"""The TIME-PARALLEL Riccati sweep (csrc/psweep.h: P wavefronts per trajectory over P + 1 horizon segments -- segment elements of the
associative LQ scan, information-form hops, the ordinary step from the true boundary values) through the batched sweep operators
(rat_dp_gain_sweep_batch / rat_dp_policy_eval_batch, switch psweep = P) against the SAME calls on the sequential sweep kernels
(psweep = 0) and against the oracle (solve_approximate_dp!, ileqg.jl:341-406; solve_approximate_dp, :412-465):
identical status / mu / Delta, gains and values to 1e-10 (measured ~1e-15)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import ratilqr.jl_amd as rat
from oracle import oracle as orc
from psweep_time import Harness, approx_of, rel
import rollprl_model

pytestmark = pytest.mark.gpu
WAVES = (2, 3, 4)            # (teams of up to four waves: one per SIMD; the eight-wave instantiations were retired in round 6)


def check(prob, x0, u, theta, waves=WAVES, mu0=None, expect_status=None, tol=1e-10):
    """Sequential kernel and every P-wave kernel against the ORACLE on every sample (gain sweep: feasibility, mu, Delta identical, L and dl
    to `tol`; policy evaluation: feasibility identical, value to `tol`), and the P-wave kernels against the sequential one."""
    B = len(theta)
    Pp, ap_o, ap = approx_of(prob, x0, u)
    hs = Harness(prob, B)
    ref = hs.gain(ap, theta, mu=mu0)
    if expect_status is not None:
        assert list(ref["st"]) == list(expect_status), ref["st"]
    okb = ref["st"] == 0
    orc_gain = [orc.dp_gain(Pp, ap_o, float(theta[b]), mu=0.0 if mu0 is None else float(mu0[b])) for b in range(B)]

    def gain_vs_oracle(g, who):
        for b, (rc, Lo, dlo, dpo, muo, deo) in enumerate(orc_gain):
            assert (rc == 0) == bool(g["st"][b] == 0), (who, b, rc, g["st"][b])
            if rc == 0:
                assert g["mu"][b] == muo and g["delta"][b] == deo, (who, b)
                assert rel(g["L"][b], Lo) < tol and rel(g["dl"][b], dlo) < tol, (who, b, rel(g["L"][b], Lo), rel(g["dl"][b], dlo))

    gain_vs_oracle(ref, "seq")
    Ls = np.where(okb[:, None, None, None], 0.9 * ref["L"], 0.0)
    mu_e = np.where(okb, np.maximum(ref["mu"], 1e-6), 1e-6)
    refe = hs.evalp(ap, Ls, theta, mu_e)
    orc_eval = [orc.dp_eval(Pp, ap_o, Ls[b], None, float(theta[b]), float(mu_e[b])) for b in range(B)]

    def eval_vs_oracle(e, who):
        for b, (rc, dpe) in enumerate(orc_eval):
            assert (rc == 0) == bool(np.isfinite(e["val"][b])), (who, b, rc, e["val"][b])
            if rc == 0:
                assert abs(e["val"][b] - dpe["s"][0]) <= tol * abs(dpe["s"][0]), (who, b, e["val"][b], dpe["s"][0])

    eval_vs_oracle(refe, "seq")
    for P in waves:
        g = hs.gain(ap, theta, mu=mu0, P=P)
        assert np.array_equal(g["st"], ref["st"]) and np.array_equal(g["mu"], ref["mu"]) and np.array_equal(g["delta"], ref["delta"]), (P, g["st"], ref["st"])
        if okb.any():
            assert rel(g["L"][okb], ref["L"][okb]) < tol and rel(g["dl"][okb], ref["dl"][okb]) < tol, P
        gain_vs_oracle(g, P)
        e = hs.evalp(ap, Ls, theta, mu_e, P=P)
        assert np.array_equal(e["st"], refe["st"]), (P, e["st"], refe["st"])
        fin = np.isfinite(refe["val"])
        assert np.array_equal(fin, np.isfinite(e["val"])) and (not fin.any() or rel(e["val"][fin], refe["val"][fin]) < tol), P
        eval_vs_oracle(e, P)
    return ref, refe


def test_headline_problem_feasible_infeasible_and_theta_zero():
    prob, x0, _ = rat.synthetic_lq_problem()
    u = 0.1 * np.random.default_rng(1).standard_normal((prob.N, prob.m))
    theta = np.array([0.0, 0.05, 0.5, 2.0, 6.0, 11.0, 12.5, 13.5, 20.0, 60.0, 1e4])          # theta = 0: the sequential body; the last ones: M not PD
    ref, refe = check(prob, x0, u, theta)
    assert np.all(ref["st"][:6] == 0) and ref["st"][-1] == 2 and np.isposinf(refe["val"][-1])


@pytest.mark.parametrize("n,m,N,kappa", [(12, 4, 50, 0.05), (4, 2, 20, 0.0), (7, 3, 33, 0.02), (12, 4, 13, 0.0), (12, 4, 5, 0.0), (3, 1, 52, 0.0)])
def test_sizes_horizons_and_cubic_drift(n, m, N, kappa):
    prob, x0, _ = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=5, kappa=kappa)
    u = 0.1 * np.random.default_rng(2).standard_normal((N, m))
    check(prob, x0, u, np.array([0.3, 1.0, 4.0, 9.0]))


def test_general_noise_covariance_and_time_varying_cost():
    """W full (not diagonal): the element step takes inv(W) G as a product instead of a row scaling; time-varying cost tables."""
    rng = np.random.default_rng(11)
    n, m, N = 12, 4, 40
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
    G = rng.standard_normal((n, n))
    W = 1e-3 * (np.eye(n) + 0.2 * (G @ G.T) / n)
    Qt = np.stack([(1.0 + 0.02 * k) * np.eye(n) for k in range(N)])
    Rt = np.stack([(0.1 + 0.005 * k) * np.eye(m) for k in range(N)])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=Qt, R=Rt, N=N, W=W, Qf=2.0 * np.eye(n))
    u = 0.1 * rng.standard_normal((N, m))
    check(prob, x0, u, np.array([0.2, 1.0, 3.0, 5.0]))


def test_time_varying_noise_covariance():
    """W(k) time-varying: every step loads its own inv(W), W and pivots; the element step and the hop are the same formulas (sweep
    operators).  The whole solve runs the time-parallel kernel's W(k) instantiations since round 6 (round 5's "wrong elements" were a
    miscompile under -amdgpu-mfma-vgpr-form: that part of kernels.hip is built without the flag; tools/wtv_probe.py)."""
    rng = np.random.default_rng(12)
    n, m, N = 9, 3, 41
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.85 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)

    def spd(k, scale):
        G = rng.standard_normal((k, k))
        return scale * (np.eye(k) + 0.2 * G @ G.T / k)

    W = np.stack([spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=spd(n, 1.0), R=spd(m, 0.2), N=N, W=W, Qf=spd(n, 1.0), kappa=0.02)
    u = 0.1 * rng.standard_normal((N, m))
    check(prob, x0, u, np.array([0.2, 1.0, 3.0, 8.0]))
    (v0, s0, i0, l0), (v1, s1, i1, l1) = _solve_both(prob, x0, u, np.array([0.0, 0.5, 2.0, 6.0, 400.0]))
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1)
    fin = np.isfinite(v0)
    assert np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-9
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, np.array([0.0, 0.5, 2.0, 6.0, 400.0]), nthreads=8)
    assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and rel(v1[fin], vo[fin]) < 1e-9


def test_mu_regularisation_restarts():
    """Indefinite stage cost: H not PD in the ordinary pass -> increase_mu_and_delta!, the whole sweep again (ileqg.jl:372-378):
    the same mu / Delta as the sequential kernel and the oracle; an element whose own recursion meets H not PD falls back."""
    rng = np.random.default_rng(1)
    n, m, N = 12, 4, 50
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
    prob = rat.LQRiskSensitiveProblem(A, B, Q=-0.2 * np.eye(n), R=0.1 * np.eye(m), N=N, W=1e-3 * np.eye(n), Qf=np.eye(n))
    ref, _ = check(prob, x0, np.zeros((N, m)), np.array([0.0, 1.0, 4.0]))
    assert np.all(ref["mu"] > 1e-6)


def test_no_state_cost_falls_back_to_the_sequential_sweep():
    """S = 0 along the whole horizon (the reference's own c = k, h = 1 test cost has no state term): the hop cannot invert S_b -- the
    workgroup's wave 0 runs the sequential body, results are its bits."""
    rng = np.random.default_rng(3)
    n, m, N = 6, 2, 30
    A, B = 0.8 * np.eye(n), rng.standard_normal((n, m))
    prob = rat.LQRiskSensitiveProblem(A, B, Q=np.zeros((n, n)), R=np.eye(m), N=N, W=1e-2 * np.eye(n), Qf=np.zeros((n, n)))
    x0, u = rng.standard_normal(n), 0.1 * rng.standard_normal((N, m))
    theta = np.array([0.5, 2.0])
    Pp, ap_o, ap = approx_of(prob, x0, u)
    hs = Harness(prob, 2)
    ref = hs.gain(ap, theta)
    for P in (2, 4):
        g = hs.gain(ap, theta, P=P)
        assert np.array_equal(g["st"], ref["st"]) and np.array_equal(g["L"], ref["L"]) and np.array_equal(g["dl"], ref["dl"])


def _solve_both(prob, x0, u, theta, opts=None, psw_kernel=True, **kw):
    """(sequential-sweep kernel, time-parallel kernel as it runs by default); for batches that leave half the device dark the default is TWO
    workgroups per sample (switch psw_duo): the one-workgroup schedule (psw_duo = 0) is held to the same bar on the way."""
    out = []
    for psw, duo in ((0, 0), (1, 0), (1, 1)):
        ctx = rat.Context(prob, opts, max_batch=len(theta), spec_eps=1, **kw)
        ctx.debug_set("block_psw", psw)
        ctx.debug_set("psw_duo", duo)
        ctx.profile(True)
        val, st, it, ls = ctx.solve_batch(x0, u, theta)
        kinds = [k for k, v in ctx.profile_get().items() if v["launches"]]
        assert kinds == ["solve_block"], kinds
        if psw and duo and psw_kernel and len(theta) <= 128 and prob.N >= 8:
            # an idle device: every sample finds its partner workgroup (observed without exception over ~50 k launches); co-residency is not
            # ASSUMED by the kernel, so only "the two-workgroup schedule ran" is asserted unless STRICT_DUO=1
            pairs = ctx.debug_get("psw_duo_count")
            assert 1 <= pairs <= len(theta) and (pairs == len(theta) or not os.environ.get("STRICT_DUO")), pairs
        else:
            assert ctx.debug_get("psw_duo_count") == 0
        out.append((val, st, it, ls))
    (v0, s0, i0, l0), (v1, s1, i1, l1), (v2, s2, i2, l2) = out
    assert np.array_equal(s1, s2) and np.array_equal(i1, i2) and np.array_equal(l1, l2), (s1, s2, i1, i2, l1, l2)
    fin = np.isfinite(v1)
    assert np.array_equal(fin, np.isfinite(v2)) and (not fin.any() or rel(v2[fin], v1[fin]) < 1e-10)
    return out[0], out[2]


@pytest.mark.parametrize("case", ["lq", "cubic", "small", "indefinite", "short"])
def test_block_solve_with_time_parallel_sweeps_equals_the_block_solve_and_the_oracle(case):
    """solve_block_psw_kernel (switch block_psw: four waves per sample, every sweep time-parallel) against solve_block_kernel and the
    oracle: identical status, iteration and line-search counts, values to 1e-9 (measured ~1e-14)."""
    opts = None
    if case == "lq":
        prob, x0, u = rat.synthetic_lq_problem()
        theta = np.concatenate([[0.0], np.linspace(0.05, 12.0, 29), [13.2, 60.0]])
    elif case == "cubic":                                                     # line searches that backtrack, > 2 iterations
        prob, x0, u = rat.synthetic_lq_problem(kappa=0.05)
        theta = np.linspace(0.0, 9.0, 24)
    elif case == "small":
        prob, x0, u = rat.synthetic_lq_problem(n=4, m=2, N=20, seed=2, kappa=0.02)
        theta = np.linspace(0.0, 3.0, 16)
    elif case == "short":                                                     # N = 9: two-wave teams at most
        prob, x0, u = rat.synthetic_lq_problem(n=6, m=2, N=9, seed=4)
        theta = np.linspace(0.0, 3.0, 8)
    else:                                                                     # mu restarts, runs to iter_max
        rng = np.random.default_rng(1)
        n, m, N = 12, 4, 50
        Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A, B, x0 = 0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
        prob = rat.LQRiskSensitiveProblem(A, B, Q=-0.2 * np.eye(n), R=0.1 * np.eye(m), N=N, W=1e-3 * np.eye(n), Qf=np.eye(n))
        u, theta, opts = np.zeros((N, m)), np.array([0.0, 1.0, 4.0]), rat.ileqg.make_opts(iter_max=6)
    (v0, s0, i0, l0), (v1, s1, i1, l1) = _solve_both(prob, x0, u, theta, opts)
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1), (s0, s1, i0, i1, l0, l1)
    fin = np.isfinite(v0)
    assert np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-9
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8, **({} if opts is None else dict(iter_max=6)))
    assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1)
    assert rel(v1[fin], vo[fin]) < 1e-9


@pytest.mark.parametrize("N", [8, 11, 17, 18, 23])
def test_short_horizons_where_the_cost_models_disagree_on_the_team_size(N):
    """N = 11 (3 / 2 waves), 17 and 18 (4 / 3): the gain-sweep and the evaluation cost models of psweep_cuts settle on different team
    sizes; the four-wave gain sweep and the four-wave evaluation share one barrier area, so the host holds them to one size (ADVICE
    r05).  Cubic drift: four iterations per solve; every solve ends in the four-wave evaluation."""
    prob, x0, u = rat.synthetic_lq_problem(n=6, m=2, N=N, seed=1, kappa=0.15)
    theta = np.linspace(0.0, 4.0, 12)
    (v0, s0, i0, l0), (v1, s1, i1, l1) = _solve_both(prob, x0, u, theta)
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1), (s0, s1, i0, i1, l0, l1)
    fin = np.isfinite(v0)
    assert fin.sum() >= 10 and i1.max() >= 3 and np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-9
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8)
    assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and rel(v1[fin], vo[fin]) < 1e-9


def test_two_wave_kernel_runs_the_last_evaluation_time_parallel():
    """257 ... 512 samples: two waves per sample (solve_block_kernel) -- the evaluation that ends the solve has the gain wave idle beside it
    and the two run it as a two-wave team (switch block_psw): identical counts, values to rounding, against block_psw = 0 and the oracle."""
    prob, x0, u = rat.synthetic_lq_problem(kappa=0.03)
    rng = np.random.default_rng(21)
    theta = np.abs(1.0 + 2.0 * rng.standard_normal(300)); theta[::41] = 0.0; theta[7] = 70.0
    (v0, s0, i0, l0), (v1, s1, i1, l1) = _solve_both(prob, x0, u, theta)
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1)
    fin = np.isfinite(v0)
    assert np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-10 and not np.array_equal(v0[fin], v1[fin])
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8)
    assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and rel(v1[fin], vo[fin]) < 1e-9


def _prl_pair(prob, x0, u, theta, opts=None, duo=1, **switches):
    """The same batch with the closed-loop rollouts of solve_block_psw_kernel on one recursion wave (psw_prl = 0) and time-parallel over the four
    waves (psw_prl = 1, the default where it applies): returns both results and the cuts the second launch used (0: it did not apply)."""
    out = []
    for prl in (0, 1):
        ctx = rat.Context(prob, opts, max_batch=len(theta), spec_eps=1)
        ctx.debug_set("psw_prl", prl)
        ctx.debug_set("psw_duo", duo)
        for k, v in switches.items():
            ctx.debug_set(k, v)
        ctx.profile(True)
        res = ctx.solve_batch(x0, u, theta)
        assert [k for k, v in ctx.profile_get().items() if v["launches"]] == ["solve_block"]
        c = ctx.debug_get("prl_cuts")
        out.append((res, (c & 0xffff, (c >> 16) & 0xffff, (c >> 32) & 0xffff)))
    assert out[0][1] == (0, 0, 0)
    return out[0][0], out[1][0], out[1][1]


@pytest.mark.parametrize("n,m,N,duo", [(12, 4, 50, 1), (12, 4, 50, 0), (12, 4, 16, 1), (12, 4, 17, 1), (7, 3, 33, 1), (3, 1, 52, 0), (12, 4, 29, 1)])
def test_time_parallel_rollout_equals_the_one_wave_rollout_and_the_oracle(n, m, N, duo):
    """rollprl_body (kappa = 0, time-invariant cost, N >= 16): the candidate's closed-loop rollout simulate_dynamics (ileqg.jl:62-87) cut into four
    segments, the deviation at each cut from the composed affine maps of the segments before it.  Counts identical, values to 1e-12 against
    the sequential recursion (VERDICT r05 item 6's bar), everything against the oracle; theta = 0, infeasible and boundary samples included."""
    prob, x0, u = rat.synthetic_lq_problem(n=n, m=m, N=N, seed=N)
    rng = np.random.default_rng(N)
    theta = np.concatenate([[0.0], np.abs(1.0 + 2.0 * rng.standard_normal(28)), [40.0, 300.0]])
    (v0, s0, i0, l0), (v1, s1, i1, l1), cuts = _prl_pair(prob, x0, u, theta, duo=duo)
    assert 0 < cuts[0] < cuts[1] < cuts[2] < N and list(cuts) == rollprl_model.cuts(N)[1:4], cuts
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1), (s0, s1, i0, i1, l0, l1)
    fin = np.isfinite(v0)
    assert fin.sum() >= 8 and np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-12
    vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8)
    assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and rel(v1[fin], vo[fin]) < 1e-9


def test_time_parallel_rollout_through_line_searches_that_backtrack_and_mu_restarts():
    """An indefinite state cost (mu restarts, solves that run to iter_max, line searches with several candidates per step: every candidate is one
    rollprl_body call with its own eps), a general noise covariance, a large initial control: the deviation form is exercised far from the
    nominal trajectory."""
    rng = np.random.default_rng(1)
    n, m, N = 12, 4, 50
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.9 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)
    G = rng.standard_normal((n, n))
    prob = rat.LQRiskSensitiveProblem(A, B, Q=-0.2 * np.eye(n), R=0.1 * np.eye(m), N=N, W=1e-3 * (np.eye(n) + 0.2 * G @ G.T / n), Qf=np.eye(n))
    opts = rat.ileqg.make_opts(iter_max=8)
    for u in (np.zeros((N, m)), 2.0 * rng.standard_normal((N, m))):
        theta = np.array([0.0, 0.3, 1.0, 2.0, 4.0, 9.0])
        (v0, s0, i0, l0), (v1, s1, i1, l1), cuts = _prl_pair(prob, x0, u, theta, opts)
        assert cuts[0] > 0 and l1.max() >= 3
        assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1), (s0, s1, i0, i1, l0, l1)
        fin = np.isfinite(v0)
        assert np.array_equal(fin, np.isfinite(v1)) and (not fin.any() or rel(v1[fin], v0[fin]) < 1e-11)
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8, iter_max=8)
        assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and (not fin.any() or rel(v1[fin], vo[fin]) < 1e-9)


def test_time_parallel_rollout_cut_models_and_where_it_does_not_apply():
    """Any cut model must give the same results (the cuts only move work between waves): extreme element / hop / terminal-tile costs, down to
    segments of one step.  Cubic drift (the deviation is not affine), a time-varying cost and horizons under 16 keep the one-wave recursion."""
    prob, x0, u = rat.synthetic_lq_problem(N=21, seed=5)
    theta = np.linspace(0.0, 6.0, 9)
    base = None
    seen = set()
    for e, h, epi in ((45, 90, 100), (1, 0, 0), (400, 0, 0), (10, 2000, 0), (45, 90, 1500), (99, 300, 300)):
        r0, r1, cuts = _prl_pair(prob, x0, u, theta, prl_elem=e, prl_hop=h, prl_epi=epi)
        assert 0 < cuts[0] < cuts[1] < cuts[2] < 21, cuts
        assert list(cuts) == rollprl_model.cuts(21, e / 100.0, h / 100.0, epi / 100.0)[1:4]      # (the host's cut model = tests/rollprl_model.py's)
        seen.add(cuts)
        base = base or r0
        for a, b in zip(r0[1:], r1[1:]):
            assert np.array_equal(a, b)
        fin = np.isfinite(base[0])
        assert np.array_equal(fin, np.isfinite(r1[0])) and rel(r1[0][fin], base[0][fin]) < 1e-12
    assert len(seen) >= 4, seen                                  # (the models above do cut differently)
    for kw in (dict(kappa=0.05), dict(N=15)):
        p2, x2, u2 = rat.synthetic_lq_problem(seed=3, **kw)
        _, _, cuts = _prl_pair(p2, x2, u2, theta)
        assert cuts == (0, 0, 0)


def test_time_parallel_rollout_in_the_noise_covariance_instantiations():
    """W(k) with a time-invariant cost and kappa = 0: the W(k) instantiations of the latency kernel (their own translation-unit part, built
    without -amdgpu-mfma-vgpr-form) run the time-parallel rollout too -- against the one-wave recursion and the oracle."""
    rng = np.random.default_rng(12)
    n, m, N = 9, 3, 41
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A, B, x0 = 0.85 * Qo, rng.standard_normal((n, m)) / np.sqrt(n), rng.standard_normal(n)

    def spd(k, scale):
        G = rng.standard_normal((k, k))
        return scale * (np.eye(k) + 0.2 * G @ G.T / k)

    W = np.stack([spd(n, 1e-3 * (0.5 + rng.random())) for _ in range(N)])
    prob = rat.LQRiskSensitiveProblem(A, B, Q=spd(n, 1.0), R=spd(m, 0.2), N=N, W=W, Qf=spd(n, 1.0), kappa=0.0)
    u = 0.1 * rng.standard_normal((N, m))
    theta = np.array([0.0, 0.5, 2.0, 6.0, 400.0])
    for duo in (1, 0):
        (v0, s0, i0, l0), (v1, s1, i1, l1), cuts = _prl_pair(prob, x0, u, theta, duo=duo)
        assert list(cuts) == rollprl_model.cuts(N)[1:4]
        assert np.array_equal(s0, s1) and np.array_equal(i0, i1) and np.array_equal(l0, l1)
        fin = np.isfinite(v0)
        assert fin.sum() >= 3 and np.array_equal(fin, np.isfinite(v1)) and rel(v1[fin], v0[fin]) < 1e-12
        vo, so, io, lo = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=8)
        assert np.array_equal(so, s1) and np.array_equal(io, i1) and np.array_equal(lo, l1) and rel(v1[fin], vo[fin]) < 1e-9
