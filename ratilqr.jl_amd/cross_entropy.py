"""Host-side mirror of ``src/cross_entropy_bilevel_optimization.jl`` (RAT iLQR) over the C ABI.

Names follow the reference's exports (src/RATiLQR.jl:36-44); ``f!`` -> ``f_``.  Randomness: the
reference draws ``rand(rng, Normal(mu, sigma))`` from a Julia MersenneTwister, which cannot be
reproduced outside Julia; here ``rng`` is either a 1-D array of standard normals consumed in order
(``theta = mu + sigma*z`` -- parity runs inject the same stream into the oracle) or an integer seed /
``numpy.random.Generator`` that seeds the library's built-in generator.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nv
from .ileqg import Context, make_opts


class CrossEntropyBilevelOptimizationSolver:
    """CrossEntropyBilevelOptimizationSolver(; kwargs...)  (cross_entropy_bilevel_optimization.jl:70-127).

    ``mu_init`` / ``sigma_init`` are mutated by ``step_`` and persist across ``solve_`` calls, as in the reference."""

    def __init__(self, mu_min_ileqg=1e-6, Delta_0_ileqg=2.0, lam_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
                 adaptive_eps_init_ileqg=False, eps_init_ileqg=1.0, eps_min_ileqg=1e-6, mu_init=1.0, sigma_init=2.0,
                 num_samples=10, num_elite=3, iter_max=5, lam=0.5, f_returns_jacobian=False, use_theta_max=False,
                 spec_eps=1, device=0):
        self.ileqg_opts = make_opts(mu_min_ileqg, Delta_0_ileqg, lam_ileqg, d_ileqg, iter_max_ileqg, eps_init_ileqg,
                                    adaptive_eps_init_ileqg, eps_min_ileqg)
        self.c = nv.CeSolver()
        nv.lib().rat_ce_default(C.byref(self.c))
        self.c.num_samples, self.c.num_elite, self.c.iter_max = int(num_samples), int(num_elite), int(iter_max)
        self.c.lam, self.c.use_theta_max = float(lam), int(bool(use_theta_max))
        self.c.mu_init, self.c.sigma_init = float(mu_init), float(sigma_init)
        self.c.mu, self.c.sigma = float(mu_init), float(sigma_init)
        self.f_returns_jacobian = f_returns_jacobian
        self.spec_eps, self.device = int(spec_eps), int(device)
        self._ctx = None
        self._rng_key = None          # what is bound to the handle of self._rng_ctx: ("seed", value) / ("gen", id) / ("stream", id)
        self._rng_ctx = None
        self._stream = None           # converted (contiguous f64) copy of the caller's stream object, made once
        self._stream_src = None
        self._stream_off = 0          # absolute position in the caller's stream of the view bound to the handle

    # mutable fields of the reference struct
    mu_init = property(lambda s: s.c.mu_init)
    sigma_init = property(lambda s: s.c.sigma_init)
    mu = property(lambda s: s.c.mu)
    sigma = property(lambda s: s.c.sigma)
    theta_max = property(lambda s: s.c.theta_max)
    theta_min = property(lambda s: s.c.theta_min)
    iter_current = property(lambda s: s.c.iter_current)
    num_samples = property(lambda s: s.c.num_samples)
    num_elite = property(lambda s: s.c.num_elite)
    iter_max = property(lambda s: s.c.iter_max)

    def context(self, problem) -> Context:
        if self._ctx is None or self._ctx.problem is not problem:
            self._ctx = Context(problem, self.ileqg_opts, max_batch=int(self.c.num_samples), spec_eps=self.spec_eps,
                                device=self.device)
        return self._ctx

    def _bind_rng(self, ctx, rng):
        """`rng` stands for the reference's stateful AbstractRNG argument: it is bound to the handle ONCE and then advances.
        * an integer seed / a numpy Generator seeds the library's generator the first time it is seen (a hand-driven loop
          ``step_(..., rng=42)`` continues the sequence instead of replaying it every CE iteration);
        * an array-like stream of N(0,1) draws is converted once and keyed on the caller's object, so the same list / view passed
          again continues where the last call stopped;
        * when the Context is recreated (another problem, a larger batch) the binding moves with it: a stream resumes at its
          current position, a seeded generator is re-seeded (its state lives in the dropped handle)."""
        L = nv.lib()
        if rng is None:
            return
        if isinstance(rng, np.random.Generator):
            key = ("gen", id(rng))
        elif isinstance(rng, (int, np.integer)):
            key = ("seed", int(rng))
        else:
            key = ("stream", id(rng))
        if key == self._rng_key and ctx is self._rng_ctx:
            return
        if key[0] == "stream":
            pos = 0
            if key == self._rng_key and self._rng_ctx is not None:          # same stream, new handle: resume where the old one stopped
                pos = self._stream_off + int(L.rat_ce_stream_pos(self._rng_ctx.h))
            if self._stream_src is not rng:
                self._stream, self._stream_src = nv.f64(rng).ravel(), rng
            z = self._stream[pos:]
            self._stream_view, self._stream_off = z, pos                    # (the view keeps the memory the handle points into alive)
            nv.check(L.rat_ce_set_stream(ctx.h, nv.P(z), C.c_int64(z.size)))
        elif key[0] == "gen":
            nv.check(L.rat_ce_seed(ctx.h, C.c_uint64(int(rng.integers(0, 2 ** 63 - 1)))))
        else:
            nv.check(L.rat_ce_seed(ctx.h, C.c_uint64(key[1])))
        self._rng_key, self._rng_ctx = key, ctx


def initialize_(ce_solver: CrossEntropyBilevelOptimizationSolver):           # initialize!  :133-138
    nv.lib().rat_ce_initialize(C.byref(ce_solver.c))


def compute_value_worker(ce_solver, problem, x, u_array, theta):             # compute_value_worker  :144-167
    """Value of one fresh iLEQG solve at ``theta``; any failure of the solve (M not PD, DomainError, ...) gives Inf (:163-165).
    The reference runs this on a worker process per sample; here it is a batch of one on the device."""
    ctx = ce_solver.context(problem)
    value, _, _, _ = ctx.solve_batch(x, u_array, np.array([float(theta)]))
    return float(value[0])


def compute_cost(ce_solver, problem, x, u_array, theta_array, kl_bound):     # compute_cost  :173-195
    theta = nv.f64(theta_array)
    if getattr(problem, "model", 0) == 0:            # closure problem: host rollouts / linearisation, the batch's sweeps in single launches
        from .generic import solve_closure_batch
        value, _, _, _ = solve_closure_batch(problem, x, u_array, theta, opts=ce_solver.ileqg_opts)
        return value + kl_bound / theta
    ctx = ce_solver.context(problem)
    if theta.size > ctx.max_batch:
        ce_solver._ctx = ctx = Context(problem, ce_solver.ileqg_opts, max_batch=theta.size, spec_eps=ce_solver.spec_eps,
                                       device=ce_solver.device)
    cost = np.zeros(theta.size)
    nv.check(nv.lib().rat_ce_compute_cost(ctx.h, nv.P(nv.f64(x)), nv.P(nv.f64(u_array)), nv.P(theta),
                                          C.c_int64(theta.size), C.c_double(kl_bound), nv.P(cost)))
    return cost


def compute_cost_serial(ce_solver, problem, x, u_array, theta_array, kl_bound):   # compute_cost_serial  :198-227
    """One solve per call (batch of 1 each), the reference's debugging twin of compute_cost."""
    theta = nv.f64(theta_array)
    assert theta.size == ce_solver.c.num_samples                                    # :204
    return np.array([compute_cost(ce_solver, problem, x, u_array, theta[i:i + 1], kl_bound)[0]
                     for i in range(theta.size)])


def get_positive_samples(mu, sigma, num_samples, rng, ce_solver=None, problem=None):   # get_positive_samples  :233-246
    """With an array ``rng`` this is pure host arithmetic; seeds need a context (ce_solver + problem)."""
    if not isinstance(rng, (int, np.integer, np.random.Generator)):
        z = nv.f64(rng)
        th = mu + sigma * z
        out = th[th > 0.0][:num_samples]
        if out.size < num_samples:
            raise nv.RatError("standard-normal stream exhausted")
        return out
    ctx = ce_solver.context(problem)
    ce_solver._bind_rng(ctx, rng)
    th = np.zeros(num_samples)
    nv.check(nv.lib().rat_ce_get_positive_samples(ctx.h, C.c_double(mu), C.c_double(sigma), C.c_int64(num_samples), nv.P(th)))
    return th


def step_(ce_solver, problem, x, u_array, kl_bound, rng, verbose=False, serial=False):   # step!  :252-335
    ctx = ce_solver.context(problem)
    ce_solver._bind_rng(ctx, rng)
    B = int(ce_solver.c.num_samples)
    th, cost = np.zeros(B), np.zeros(B)
    nv.check(nv.lib().rat_ce_step(ctx.h, C.byref(ce_solver.c), nv.P(nv.f64(x)), nv.P(nv.f64(u_array)),
                                  C.c_double(kl_bound), nv.P(th), nv.P(cost)))
    return th, cost


def solve_(ce_solver, problem, x_0, u_array, rng, kl_bound, verbose=False, serial=False):   # solve!  :364-415
    """Returns (θ_opt, x_array, l_array, L_array, value, θ_min, θ_max)."""
    assert kl_bound >= 0, "KL Divergence Bound must be non-negative"
    ctx = ce_solver.context(problem)
    ce_solver._bind_rng(ctx, rng)
    n, m, N = ctx.n, ctx.m, ctx.N
    x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
    th, val, tmin, tmax = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    nv.check(nv.lib().rat_ce_solve(ctx.h, C.byref(ce_solver.c), nv.P(nv.f64(x_0)), nv.P(nv.f64(u_array)),
                                   C.c_double(kl_bound), C.byref(th), nv.P(x), nv.P(l), nv.P(Lb), C.byref(val),
                                   C.byref(tmin), C.byref(tmax)))
    return th.value, x, l, nv.from_cm3(Lb, N, m, n), val.value, tmin.value, tmax.value
