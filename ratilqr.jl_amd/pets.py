"""Host-side mirror of ``src/pets.jl`` (PETS: cross-entropy over control sequences with stochastic rollouts).

Names follow the reference's exports (src/RATiLQR.jl:55-62); ``f!`` -> ``f_``.  Array conventions: a control sequence array
is ``[sample, t, a]``; ``mu_array`` is ``[t, a]``; ``Sigma_array`` is ``[t, a, b]``.  ``rng`` is a ``numpy.random.Generator``;
the mirror draws the standard-normal / uniform streams from it in the reference's serial order and injects them, so a run is
reproducible from the generator state (the Julia MersenneTwister stream itself cannot be reproduced)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nv
from .ileqg import Context


def _cm3(a):
    return np.ascontiguousarray(np.asarray(a, float).transpose(0, 2, 1)).ravel()


class CrossEntropyDirectOptimizationSolver:                       # a.k.a. "PETS", pets.jl:36-68
    def __init__(self, mu_init_array, Sigma_init_array, num_control_samples=10, num_trajectory_samples=10, num_elite=3,
                 iter_max=5, smoothing_factor=0.1, device=0):
        mu_init_array, Sigma_init_array = np.asarray(mu_init_array, float), np.asarray(Sigma_init_array, float)
        assert len(mu_init_array) == len(Sigma_init_array)
        self.N, self.m = mu_init_array.shape
        self._mu_init, self._Sig_init = nv.f64(mu_init_array).ravel().copy(), _cm3(Sigma_init_array).copy()
        self._mu, self._Sig = self._mu_init.copy(), self._Sig_init.copy()
        self.c = nv.PetsSolver()
        self.c.num_control_samples, self.c.num_trajectory_samples = int(num_control_samples), int(num_trajectory_samples)
        self.c.num_elite, self.c.iter_max, self.c.smoothing_factor = int(num_elite), int(iter_max), float(smoothing_factor)
        self.c.N, self.c.m, self.c.iter_current = self.N, self.m, 0
        self.c.mu_init, self.c.Sigma_init, self.c.mu, self.c.Sigma = nv.P(self._mu_init), nv.P(self._Sig_init), nv.P(self._mu), nv.P(self._Sig)
        self.device = int(device)
        self._ctx = None

    num_control_samples = property(lambda s: s.c.num_control_samples)
    num_trajectory_samples = property(lambda s: s.c.num_trajectory_samples)
    num_elite = property(lambda s: s.c.num_elite)
    iter_max = property(lambda s: s.c.iter_max)
    smoothing_factor = property(lambda s: s.c.smoothing_factor)

    @property
    def iter_current(self):
        return self.c.iter_current

    @iter_current.setter
    def iter_current(self, v):
        self.c.iter_current = int(v)

    @property
    def mu_array(self):
        return self._mu.reshape(self.N, self.m).copy()

    @mu_array.setter
    def mu_array(self, v):
        self._mu[:] = nv.f64(v).ravel()

    @property
    def Sigma_array(self):
        return self._Sig.reshape(self.N, self.m, self.m).transpose(0, 2, 1).copy()

    @Sigma_array.setter
    def Sigma_array(self, v):
        self._Sig[:] = _cm3(v)

    mu_init_array = property(lambda s: s._mu_init.reshape(s.N, s.m).copy())
    Sigma_init_array = property(lambda s: s._Sig_init.reshape(s.N, s.m, s.m).transpose(0, 2, 1).copy())

    def context(self, problem):
        if self._ctx is None or self._ctx[0] is not problem:
            ctx = Context(problem.lq, device=self.device)
            g, keep = make_gen_desc(problem)
            nv.check(nv.lib().rat_pets_problem_set(ctx.h, C.byref(g)))
            self._ctx = (problem, ctx, keep)
        return self._ctx[1]


def make_gen_desc(problem):
    """rat_gen_problem_desc of an LQGenerativeProblem (+ the arrays it points into, to be kept alive)."""
    t = problem.gen_tables()
    desc, keep = nv.make_desc(problem.lq)
    g = nv.GenProblemDesc()
    g.lq = desc
    arrs = {k: nv.f64(t[k]) for k in ("nmean", "nchol", "tmean2", "tchol2")}
    g.l1u, g.noise_kind, g.nlo, g.nhi, g.tw2 = t["l1u"], t["noise_kind"], t["nlo"], t["nhi"], t["tw2"]
    g.nmean, g.nchol, g.tmean2, g.tchol2 = (nv.P(arrs[k]) for k in ("nmean", "nchol", "tmean2", "tchol2"))
    return g, (keep, arrs)


def initialize_(direct_solver):                                   # initialize!  pets.jl:70-74
    nv.lib().rat_pets_initialize(C.byref(direct_solver.c))


def draw_noise(problem, rng, S, K, use_true_model=False):
    """The N(0,1)/U[0,1) draws compute_cost_serial consumes, in its order (sample, trajectory, step, component)."""
    n, N = problem.n, problem.N
    if use_true_model and problem.tw2 > 0:
        zu = rng.random(S * K * N)
        zn = rng.standard_normal(S * K * N * n)
    else:
        zu = None
        zn = rng.random(S * K * N * n) if problem.noise_kind == 1 else rng.standard_normal(S * K * N * n)
    return zn, zu


def compute_cost_serial(direct_solver, problem, x, control_sequence_array, rng, use_true_model=False, streams=None, seed=None):
    """pets.jl:128-157.  ``streams=(zn, zu)`` injects the draws; ``seed`` selects the device generator instead."""
    ctrl = nv.f64(control_sequence_array)
    S = ctrl.shape[0]
    assert S == direct_solver.c.num_control_samples and ctrl.shape[1] == direct_solver.N
    K = int(direct_solver.c.num_trajectory_samples)
    ctx = direct_solver.context(problem)
    zn = zu = None
    if seed is None:
        zn, zu = streams if streams is not None else draw_noise(problem, rng, S, K, use_true_model)
        zn = nv.f64(zn)
        zu = None if zu is None else nv.f64(zu)
    cost = np.zeros(S)
    nv.check(nv.lib().rat_pets_compute_cost(ctx.h, nv.P(nv.f64(x)), nv.P(ctrl), C.c_int64(S), C.c_int64(K), int(use_true_model),
                                            nv.P(zn), nv.P(zu), C.c_uint64(0 if seed is None else int(seed)), nv.P(cost)))
    return cost


compute_cost = compute_cost_serial            # the reference's distributed twin differs only in its RNG jumps (pets.jl:100-126)


def compute_cost_worker(direct_solver, problem, x, u_array, rng, use_true_model=False, streams=None, seed=None):   # pets.jl:76-98
    """Mean cost of ONE control sequence over num_trajectory_samples stochastic rollouts (what a worker computes per sample)."""
    ctrl = nv.f64(u_array)[None]
    assert ctrl.shape[1] == direct_solver.N
    K = int(direct_solver.c.num_trajectory_samples)
    ctx = direct_solver.context(problem)
    zn = zu = None
    if seed is None:
        zn, zu = streams if streams is not None else draw_noise(problem, rng, 1, K, use_true_model)
        zn = nv.f64(zn)
        zu = None if zu is None else nv.f64(zu)
    cost = np.zeros(1)
    nv.check(nv.lib().rat_pets_compute_cost(ctx.h, nv.P(nv.f64(x)), nv.P(ctrl), C.c_int64(1), C.c_int64(K), int(use_true_model),
                                            nv.P(zn), nv.P(zu), C.c_uint64(0 if seed is None else int(seed)), nv.P(cost)))
    return float(cost[0])


def get_elite_samples(direct_solver, control_sequence_array, cost_array):          # pets.jl:159-171
    ctrl = nv.f64(control_sequence_array)
    order = np.argsort(nv.f64(cost_array), kind="stable")[: direct_solver.c.num_elite]
    return ctrl[order]


def compute_new_distribution(direct_solver, control_sequence_elite_array):          # pets.jl:173-191
    el = nv.f64(control_sequence_elite_array)
    sf = direct_solver.c.smoothing_factor
    mean, var = el.mean(axis=0), el.var(axis=0, ddof=1)
    mu_new = (1.0 - sf) * mean + sf * direct_solver.mu_array
    Sig_new = (1.0 - sf) * np.stack([np.diag(v) for v in var]) + sf * direct_solver.Sigma_array
    return mu_new, Sig_new


def step_(direct_solver, problem, x, rng, use_true_model=False, verbose=False, serial=True, seed=None):   # step!  pets.jl:193-245
    ctx = direct_solver.context(problem)
    S, K, N, m = (int(direct_solver.c.num_control_samples), int(direct_solver.c.num_trajectory_samples), direct_solver.N, direct_solver.m)
    zc = nv.f64(rng.standard_normal(S * N * m))
    zn = zu = None
    if seed is None:
        zn, zu = draw_noise(problem, rng, S, K, use_true_model)
        zn, zu = nv.f64(zn), (None if zu is None else nv.f64(zu))
    ctrl, cost = np.zeros((S, N, m)), np.zeros(S)
    nv.check(nv.lib().rat_pets_step(ctx.h, C.byref(direct_solver.c), nv.P(nv.f64(x)), int(use_true_model), nv.P(zc), nv.P(zn), nv.P(zu),
                                    C.c_uint64(0 if seed is None else int(seed)), nv.P(ctrl), nv.P(cost)))
    return ctrl, cost


def solve_(direct_solver, problem, x_0, rng, use_true_model=False, verbose=False, serial=True, seed=None):   # solve!  pets.jl:270-281
    """Returns (mu_array, Sigma_array).  With a `seed` (rollout noise from the device generator) the whole loop is one library call
    (rat_pets_solve: sampling, rollouts, elites and the smoothed update stay on the device, one host wait); the control normals are the
    ones step_ would draw from `rng`, iteration after iteration.  rng = None: they are drawn on the device as well."""
    if seed is not None:
        ctx = direct_solver.context(problem)
        c = direct_solver.c
        S, N, m = int(c.num_control_samples), direct_solver.N, direct_solver.m
        zc = None if rng is None else nv.f64(rng.standard_normal(int(c.iter_max) * S * N * m))
        nv.check(nv.lib().rat_pets_solve(ctx.h, C.byref(c), nv.P(nv.f64(x_0)), int(use_true_model), nv.P(zc), None, None, C.c_uint64(int(seed))))
        return direct_solver.mu_array, direct_solver.Sigma_array
    initialize_(direct_solver)
    while direct_solver.c.iter_current < direct_solver.c.iter_max:
        step_(direct_solver, problem, x_0, rng, use_true_model, verbose, serial,
              seed=None if seed is None else seed + direct_solver.c.iter_current)
    return direct_solver.mu_array, direct_solver.Sigma_array
