"""GPU restatement of /root/reference/test/pets_test.jl through the C ABI, and parity of the PETS rollout-cost kernel with
the oracle on injected noise streams (serial semantics of compute_cost_serial, pets.jl:128-157).  Tolerance 1e-11 relative
(sums of 20-100 rollout costs; FMA contraction differs from the CPU).  The device generator (Philox) is checked statistically."""
import numpy as np
import pytest

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import pets
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
N = 20


def ref_problem():                                            # pets_test.jl:15-20
    return rat.LQGenerativeProblem(np.eye(2), np.eye(2), N, ("uniform", 0.0, 1.0), l1u=1.0, q0f=1.0)


def test_reference_pets_test():                               # pets_test.jl:22-94
    prob = ref_problem()
    mu0, Sig0 = np.zeros((N, 2)), np.stack([np.eye(2)] * N)
    ds = rat.CrossEntropyDirectOptimizationSolver(mu0, Sig0, num_control_samples=20, num_trajectory_samples=100, num_elite=5,
                                                  iter_max=20, smoothing_factor=0.1)
    assert ds.N == N and ds.iter_current == 0 and np.all(ds.mu_array == mu0) and np.all(ds.Sigma_array == Sig0)   # :30-33
    ds.iter_current = 10
    ds.mu_array = np.ones((N, 2))
    ds.Sigma_array = np.stack([0.1 * np.eye(2)] * N)
    pets.initialize_(ds)
    assert ds.iter_current == 0 and np.all(ds.mu_array == mu0) and np.all(ds.Sigma_array == Sig0)                 # :35-41
    rng = np.random.default_rng(1234)
    ctrl = rng.random((20, N, 2))
    x_init = np.zeros(2)
    zn, _ = pets.draw_noise(prob, rng, 20, 100)
    cost = pets.compute_cost_serial(ds, prob, x_init, ctrl, None, streams=(zn, None))
    cost2 = pets.compute_cost(ds, prob, x_init, ctrl, None, streams=(zn, None))
    assert np.all(cost == cost2) and cost.size == 20                                                              # :47-53
    for ii in range(20):                                                                                          # :54-63
        x, c = x_init, 0.0
        for t in range(N):
            c += prob.c(t, x, ctrl[ii, t])
            x = x + ctrl[ii, t] + zn[(ii * 100 * N + t) * 2:(ii * 100 * N + t) * 2 + 2]
        c += prob.h(x)
        assert np.isclose(c, cost[ii], rtol=1e-12)
    elite = pets.get_elite_samples(ds, ctrl, cost)                                                                # :66-70
    assert len(elite) == 5 and np.array_equal(elite, ctrl[np.argsort(cost, kind="stable")[:5]])
    mu_new, Sig_new = pets.compute_new_distribution(ds, elite)                                                    # :73-84
    assert mu_new.shape == (N, 2) and Sig_new.shape == (N, 2, 2)
    for t in range(N):
        assert np.allclose(mu_new[t], 0.9 * elite[:, t].mean(axis=0) + 0.1 * ds.mu_array[t])
        assert np.allclose(Sig_new[t], 0.9 * np.diag(elite[:, t].var(axis=0, ddof=1)) + 0.1 * ds.Sigma_array[t])
    pets.step_(ds, prob, x_init, np.random.default_rng(1234))                                                     # :87-89
    assert ds.iter_current == 1
    mu, Sig = pets.solve_(ds, prob, x_init, rng)                                                                  # :92-94
    assert ds.iter_current == ds.iter_max
    assert np.abs(mu).mean() < 0.25 and Sig.max() < 0.5       # c = sum|u|: the CE distribution contracts towards u = 0


def test_compute_cost_worker_is_one_row_of_compute_cost():    # pets.jl:76-98
    prob = ref_problem()
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((N, 2)), np.stack([np.eye(2)] * N), num_control_samples=4, num_trajectory_samples=50)
    rng = np.random.default_rng(5)
    ctrl = rng.random((4, N, 2))
    zn, _ = pets.draw_noise(prob, rng, 4, 50)
    cost = pets.compute_cost_serial(ds, prob, np.zeros(2), ctrl, None, streams=(zn, None))
    per = N * 2 * 50
    for ii in range(4):
        assert pets.compute_cost_worker(ds, prob, np.zeros(2), ctrl[ii], None, streams=(zn[ii * per:(ii + 1) * per], None)) == cost[ii]


def rich_problem(n=12, m=4, Nh=30, kappa=-0.01):
    r = np.random.default_rng(8)
    A = 0.9 * np.linalg.qr(r.standard_normal((n, n)))[0]
    cov = 0.02 * np.eye(n) + 0.01 * np.outer(np.ones(n), np.ones(n)) / n
    return rat.LQGenerativeProblem(A, r.standard_normal((n, m)) / np.sqrt(n), Nh, ("gaussian", 0.05 * r.standard_normal(n), cov),
                                   Q=np.eye(n), R=0.1 * np.eye(m), P=0.02 * r.standard_normal((m, n)), qv=0.1 * r.standard_normal(n),
                                   rv=0.1 * r.standard_normal(m), q0=0.5, Qf=2 * np.eye(n), qvf=0.1 * r.standard_normal(n), q0f=1.0,
                                   kappa=kappa, l1u=0.2, true_noise=(0.3, 0.2 * np.ones(n), 0.05 * np.eye(n))), r


@pytest.mark.parametrize("use_true", [False, True])
def test_rollout_costs_match_oracle_on_injected_streams(use_true):
    prob, r = rich_problem()
    S, K = 24, 50
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([np.eye(4)] * 30), num_control_samples=S, num_trajectory_samples=K)
    ctrl = 0.3 * r.standard_normal((S, 30, 4))
    x0 = r.standard_normal(12)
    zn, zu = r.standard_normal(S * K * 30 * 12), r.random(S * K * 30)
    got = pets.compute_cost_serial(ds, prob, x0, ctrl, None, use_true, streams=(zn, zu if use_true else None))
    ref = orc.pets_compute_cost(orc.GenProblem(prob), x0, ctrl, K, use_true, zn, zu if use_true else None)
    assert np.all(np.isfinite(ref)) and np.all(np.abs(got - ref) <= 1e-11 * np.abs(ref))


def test_step_and_solve_match_oracle():
    prob, r = rich_problem(n=6, m=2, Nh=15, kappa=0.0)      # wide control samples: keep the dynamics linear (no blow-up)
    S, K, Nh, m, n = 16, 20, 15, 2, 6
    kw = dict(num_control_samples=S, num_trajectory_samples=K, num_elite=4, iter_max=3, smoothing_factor=0.2)
    mu0, Sig0 = 0.1 * np.ones((Nh, m)), np.stack([np.array([[1.0, 0.3], [0.3, 0.5]])] * Nh)
    ds = rat.CrossEntropyDirectOptimizationSolver(mu0, Sig0, **kw)
    so = orc.PetsSolver(mu0, Sig0, **kw)
    G = orc.GenProblem(prob)
    x0 = r.standard_normal(n)
    rng_g, rng_o = np.random.default_rng(77), np.random.default_rng(77)
    for it in range(3):
        ctrl_g, cost_g = pets.step_(ds, prob, x0, rng_g)
        zc = rng_o.standard_normal(S * Nh * m)
        zn = rng_o.standard_normal(S * K * Nh * n)
        rc, ctrl_o, cost_o = so.step(G, x0, False, zc, zn)
        assert rc == 0 and np.allclose(ctrl_g, ctrl_o, rtol=1e-13, atol=1e-15)
        assert np.all(np.isfinite(cost_o)) and np.all(np.abs(cost_g - cost_o) <= 1e-11 * np.abs(cost_o))
        assert np.allclose(ds.mu_array, so.mu_array, rtol=1e-12) and np.allclose(ds.Sigma_array, so.Sigma_array, rtol=1e-12, atol=1e-300)
    assert ds.iter_current == 3


@pytest.mark.parametrize("use_true", [False, True])
@pytest.mark.parametrize("S,K,Nh", [(24, 50, 30), (7, 13, 30), (3, 16, 7), (1, 1, 30), (90, 400, 30)])
def test_sixteen_per_wave_kernels_equal_four_per_wave(S, K, Nh, use_true):
    """The rollout kernels -- 16 trajectories per wavefront as MFMA columns with the noise drawn by the recursion's own wavefront (switch
    pets_wave16 = 2) or by three generator wavefronts of its workgroup (3; what small launches use under the default 1), and 4 per
    wavefront (0) -- evaluate the same sums: 2 and 3 bit for bit, 0 to rounding, on injected streams and on the device generator too
    (all key Philox by (trajectory, step pair, component), so a seed names the same noise in each).  K not a multiple of 16 leaves dead
    columns; an odd horizon leaves half a step pair."""
    prob, r = rich_problem(Nh=Nh)
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, 4)), np.stack([np.eye(4)] * Nh), num_control_samples=S, num_trajectory_samples=K)
    ctrl = 0.3 * r.standard_normal((S, Nh, 4))
    x0 = r.standard_normal(12)
    inject = S * K <= 2000
    zn, zu = (r.standard_normal(S * K * Nh * 12), r.random(S * K * Nh)) if inject else (None, None)
    streams = (zn, zu if use_true else None)
    ctx = ds.context(prob)
    assert ctx.debug_get("pets_wave16") == 1
    got, gen = {}, {}
    for mode in (1, 2, 3, 0):
        ctx.debug_set("pets_wave16", mode)
        if inject:
            got[mode] = pets.compute_cost_serial(ds, prob, x0, ctrl, None, use_true, streams=streams)
        gen[mode] = pets.compute_cost_serial(ds, prob, x0, ctrl, None, use_true, seed=9)
    ctx.debug_set("pets_wave16", 1)
    if inject:
        ref = orc.pets_compute_cost(orc.GenProblem(prob), x0, ctrl, K, use_true, zn, zu if use_true else None)
        assert np.all(np.isfinite(ref)) and np.all(np.abs(got[1] - ref) <= 1e-11 * np.abs(ref))
        assert np.array_equal(got[2], got[3]) and np.array_equal(got[1], got[3])
        assert np.all(np.abs(got[2] - got[0]) <= 1e-12 * np.abs(got[0]))
    assert np.all(np.isfinite(gen[0])) and np.array_equal(gen[2], gen[3]) and np.array_equal(gen[1], gen[2])
    assert np.all(np.abs(gen[2] - gen[0]) <= 1e-12 * np.abs(gen[0]))


def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on numpy arrays of 32-bit words held in uint64, as csrc/kernels.hip:philox4x32_10 writes it."""
    M = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) & M for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & M, np.uint64(k1) & M
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


@pytest.mark.parametrize("mode", [1, 2, 3, 0])
def test_device_generator_against_a_host_restatement(mode):
    """The device noise itself, end to end: with x_{t+1} = w_t (A = B = 0), w ~ N(0, I) and cost 1/2 |x|^2 a trajectory's cost is half the sum
    of the squares of its N x 12 standard normals.  The host draws them the way the kernels document -- Philox4x32-10 keyed by the seed,
    counter (trajectory, step pair, component), two 53-bit uniforms per block, Box-Muller by csrc/rat_normal.h (oracle/normal_check.c) giving
    the normals of steps 2p and 2p + 1 -- and must reproduce every sample cost.  An odd horizon leaves half a pair unused."""
    import ctypes as C
    from test_cpu_normal import parts
    n, m, Nh, S, K, seed = 12, 4, 7, 5, 37, 0x1234567890ABCDEF
    prob = rat.LQGenerativeProblem(np.zeros((n, n)), np.zeros((n, m)), Nh, ("gaussian", np.zeros(n), np.eye(n)), Q=np.eye(n), R=np.eye(m),
                                   Qf=np.eye(n), kappa=0.0)
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((Nh, m)), np.stack([np.eye(m)] * Nh), num_control_samples=S, num_trajectory_samples=K)
    ds.context(prob).debug_set("pets_wave16", mode)
    got = pets.compute_cost_serial(ds, prob, np.zeros(n), np.zeros((S, Nh, m)), None, seed=seed)
    tj, p, c = np.meshgrid(np.arange(S * K), np.arange((Nh + 1) // 2), np.arange(n), indexing="ij")
    r0, r1, r2, r3 = _philox4x32_10(tj, 0 * tj, p, c, seed & 0xFFFFFFFF, seed >> 32)
    u1 = (((r0 << np.uint64(32)) | r1) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    u2 = (((r2 << np.uint64(32)) | r3) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    _, _, _, _, z0, z1 = parts(np.ascontiguousarray(u1.ravel()), np.ascontiguousarray(u2.ravel()))
    z = np.stack([z0.reshape(u1.shape), z1.reshape(u1.shape)], axis=2).reshape(S * K, -1, n)[:, :Nh]      # [trajectory][step][component]
    ref = (0.5 * (z ** 2).sum(axis=(1, 2))).reshape(S, K).mean(axis=1)
    assert np.all(np.abs(got - ref) <= 1e-12 * ref)


def test_device_generator_is_statistically_sane():
    """BASELINE config 5 shape: 10k trajectory samples, N = 30, device-generated noise (Philox): mean cost within a few
    standard errors of the injected-stream evaluation; reproducible for a fixed seed, different across seeds."""
    prob, r = rich_problem()
    S, K = 100, 100
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([np.eye(4)] * 30), num_control_samples=S, num_trajectory_samples=K)
    ctrl = 0.2 * r.standard_normal((S, 30, 4))
    x0 = r.standard_normal(12)
    c1 = pets.compute_cost_serial(ds, prob, x0, ctrl, None, seed=42)
    c1b = pets.compute_cost_serial(ds, prob, x0, ctrl, None, seed=42)
    c2 = pets.compute_cost_serial(ds, prob, x0, ctrl, None, seed=43)
    cref = pets.compute_cost_serial(ds, prob, x0, ctrl, np.random.default_rng(1))
    assert np.array_equal(c1, c1b) and not np.array_equal(c1, c2)
    se = np.std(c1 - c2) / np.sqrt(2)                      # per-sample Monte-Carlo standard error of a K-rollout mean
    assert abs(np.mean(c1 - cref)) < 5 * se / np.sqrt(S) * np.sqrt(2) + 1e-9
    assert abs(np.mean(c1) / np.mean(cref) - 1) < 0.02


def test_config5_ten_thousand_trajectories():
    """BASELINE config 5: 10k stochastic trajectories (100 control samples x 100 rollouts), N = 30, n = 12, m = 4, one call.
    Injected noise: every sample's mean cost equals the oracle's; device noise: finite, reproducible, and a Monte-Carlo estimate of the
    same mean (the K = 100 rollouts of a sample average out to within a few standard errors)."""
    prob, r = rich_problem()
    S, K = 100, 100
    ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([np.eye(4)] * 30), num_control_samples=S, num_trajectory_samples=K)
    ctrl = 0.3 * r.standard_normal((S, 30, 4))
    x0 = r.standard_normal(12)
    zn = r.standard_normal(S * K * 30 * 12)
    got = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, streams=(zn, None))
    ref = orc.pets_compute_cost(orc.GenProblem(prob), x0, ctrl, K, False, zn, None)
    assert np.all(np.isfinite(ref)) and np.all(np.abs(got - ref) <= 1e-11 * np.abs(ref))
    dev = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11)
    dev2 = pets.compute_cost_serial(ds, prob, x0, ctrl, None, False, seed=11)
    assert np.all(np.isfinite(dev)) and np.array_equal(dev, dev2)
    assert np.median(np.abs(dev - got) / np.abs(got)) < 0.05          # two independent K = 100 estimates of the same means


@pytest.mark.parametrize("S,K,Nh,n,m,ne,use_true", [(100, 100, 30, 12, 4, 10, False), (16, 20, 15, 6, 2, 4, False), (37, 33, 30, 12, 4, 5, True),
                                                     (1024, 4, 9, 12, 3, 100, False), (2, 7, 30, 12, 4, 2, False),
                                                     (12, 6, 600, 6, 2, 3, False)])       # N = 600: > 64 KB of LDS in the bookkeeping launch
def test_device_resident_solve_equals_the_host_loop(S, K, Nh, n, m, ne, use_true):
    """rat_pets_solve with the loop over control sequences on the device (ce_device.hip: pets_sample_kernel / pets_update_kernel, ONE host
    wait per solve!) against the same call with the switch pets_device = 0 (sample / update on the host between device calls,
    pets.jl:159-245): mu and Sigma bit for bit on an injected stream of control normals, rollout noise from the device generator."""
    prob, r = rich_problem(n=n, m=m, Nh=Nh, kappa=0.0)
    mu0 = 0.05 * r.standard_normal((Nh, m))
    Sig0 = np.stack([0.3 * np.eye(m) + 0.05 * np.ones((m, m))] * Nh)
    x0 = r.standard_normal(n)
    kw = dict(num_control_samples=S, num_trajectory_samples=K, num_elite=ne, iter_max=4, smoothing_factor=0.15)
    out = []
    for dev in (1, 0):
        ds = rat.CrossEntropyDirectOptimizationSolver(mu0, Sig0, **kw)
        ctx = ds.context(prob)
        ctx.debug_set("pets_device", dev)
        ctx.profile(True); ctx.profile_reset()
        mu, Sig = pets.solve_(ds, prob, x0, np.random.default_rng(5), use_true_model=use_true, seed=1234)
        kinds = {k: v["launches"] for k, v in ctx.profile_get().items() if v["launches"]}
        ctx.profile(False)
        assert ds.iter_current == 4
        out.append((mu.copy(), Sig.copy(), kinds))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2].get("ce_bookkeeping") == 5 and "ce_bookkeeping" not in out[1][2]          # sample | (update + sample) x 3 | update: on the device
    assert np.all(np.isfinite(out[0][0])) and not np.array_equal(out[0][0], mu0)


def test_device_resident_solve_draws_its_own_control_normals_and_reports_a_bad_covariance():
    prob, r = rich_problem(kappa=0.0)
    Nh, m = 30, 4
    mu0, Sig0 = np.zeros((Nh, m)), np.stack([0.25 * np.eye(m)] * Nh)
    x0 = r.standard_normal(12)
    ds = rat.CrossEntropyDirectOptimizationSolver(mu0, Sig0, num_control_samples=64, num_trajectory_samples=32, num_elite=8, iter_max=3)
    mu_a, Sig_a = pets.solve_(ds, prob, x0, None, seed=7)                   # rng = None: Philox on the device for the control normals too
    mu_a, Sig_a = mu_a.copy(), Sig_a.copy()
    mu_b, Sig_b = pets.solve_(ds, prob, x0, None, seed=7)
    assert np.array_equal(mu_a, mu_b) and np.array_equal(Sig_a, Sig_b) and np.all(np.isfinite(mu_a))          # a seed names the draws
    mu_c, _ = pets.solve_(ds, prob, x0, None, seed=8)
    assert not np.array_equal(mu_a, mu_c)
    dg = np.diagonal(Sig_a, axis1=1, axis2=2)
    assert np.all(dg > 0) and np.all(np.isfinite(dg)) and np.count_nonzero(Sig_a) == dg.size               # Diagonal(var) + smoothing: stays diagonal
    bad = Sig0.copy(); bad[3] = -np.eye(m)
    ds_bad = rat.CrossEntropyDirectOptimizationSolver(mu0, bad, num_control_samples=8, num_trajectory_samples=4, num_elite=2, iter_max=2)
    with pytest.raises(rat.RatError):
        pets.solve_(ds_bad, prob, x0, np.random.default_rng(1), seed=3)
