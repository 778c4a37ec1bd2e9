"""Host-side mirror of ``src/nelder_mead_bilevel_optimization.jl`` (RAT iLQR++) over the C ABI.

Names follow the reference's exports (src/RATiLQR.jl:45-49); ``f!`` -> ``f_``.  Every Nelder-Mead iteration is ONE
batched device call (the <= 6 vertices the sequential logic can ask for), see ``rat_nm_step`` in include/ratilqr.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nv
from .ileqg import Context, make_opts


class NelderMeadBilevelOptimizationSolver:
    """NelderMeadBilevelOptimizationSolver(; kwargs...)  (nelder_mead_bilevel_optimization.jl:72-128).

    ``c_high`` / ``c_low`` (``None`` until first computed) and the possibly shrunk ``theta_*_init`` persist across
    ``solve_`` calls exactly as in the reference (its ``initialize!`` does not reset them)."""

    def __init__(self, mu_min_ileqg=1e-6, Delta_0_ileqg=2.0, lam_ileqg=0.5, d_ileqg=1e-2, iter_max_ileqg=100,
                 adaptive_eps_init_ileqg=False, eps_init_ileqg=1.0, eps_min_ileqg=1e-6, f_returns_jacobian=False,
                 alpha=1.0, beta=2.0, gamma=0.5, eps=1e-2, lam=0.5, iter_max=100, theta_high_init=3.0, theta_low_init=1e-8,
                 device=0):
        self.ileqg_opts = make_opts(mu_min_ileqg, Delta_0_ileqg, lam_ileqg, d_ileqg, iter_max_ileqg, eps_init_ileqg,
                                    adaptive_eps_init_ileqg, eps_min_ileqg)
        self.c = nv.NmSolver()
        nv.lib().rat_nm_default(C.byref(self.c))
        self.c.alpha, self.c.beta, self.c.gamma, self.c.eps, self.c.lam = alpha, beta, gamma, eps, lam
        self.c.iter_max = int(iter_max)
        self.c.theta_high_init = self.c.theta_high = float(theta_high_init)
        self.c.theta_low_init = self.c.theta_low = float(theta_low_init)
        self.f_returns_jacobian = f_returns_jacobian
        self.device = int(device)
        self._ctx = None

    theta_high = property(lambda s: s.c.theta_high)
    theta_low = property(lambda s: s.c.theta_low)
    theta_high_init = property(lambda s: s.c.theta_high_init)
    theta_low_init = property(lambda s: s.c.theta_low_init)
    iter_current = property(lambda s: s.c.iter_current)
    c_high = property(lambda s: s.c.c_high if s.c.has_c_high else None)
    c_low = property(lambda s: s.c.c_low if s.c.has_c_low else None)

    def context(self, problem) -> Context:
        if self._ctx is None or self._ctx.problem is not problem:
            # 1024 samples: both initial vertices and three iterations' worth of vertices in the first device call (driver.cpp: rat_nm_solve)
            self._ctx = Context(problem, self.ileqg_opts, max_batch=1024, spec_eps=1, device=self.device)
        return self._ctx


def initialize_(nm_solver):                                                    # initialize!  :164-168
    nv.lib().rat_nm_initialize(C.byref(nm_solver.c))


def compute_cost_worker(nm_solver, problem, x, u_array, theta, kl_bound):      # :134-158
    ctx = nm_solver.context(problem)
    out = C.c_double()
    nv.check(nv.lib().rat_nm_compute_cost(ctx.h, nv.P(nv.f64(x)), nv.P(nv.f64(u_array)), C.c_double(theta),
                                          C.c_double(kl_bound), C.byref(out)))
    return out.value


def step_(nm_solver, problem, x, u_array, kl_bound, verbose=False):            # step!  :174-252
    ctx = nm_solver.context(problem)
    nv.check(nv.lib().rat_nm_step(ctx.h, C.byref(nm_solver.c), nv.P(nv.f64(x)), nv.P(nv.f64(u_array)), C.c_double(kl_bound)))


def solve_(nm_solver, problem, x_0, u_array, kl_bound, verbose=False):         # solve!  :276-352
    """Returns (θ_opt, x_array, l_array, L_array, value)."""
    assert kl_bound >= 0, "KL Divergence Bound must be non-negative"
    ctx = nm_solver.context(problem)
    n, m, N = ctx.n, ctx.m, ctx.N
    x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
    th, val, st = C.c_double(), C.c_double(), C.c_int32()
    nv.check(nv.lib().rat_nm_solve(ctx.h, C.byref(nm_solver.c), nv.P(nv.f64(x_0)), nv.P(nv.f64(u_array)), C.c_double(kl_bound),
                                   C.byref(th), nv.P(x), nv.P(l), nv.P(Lb), C.byref(val), C.byref(st)))
    if st.value in (nv.ST_M_NOT_PD_INIT, nv.ST_M_NOT_PD_GAIN):
        raise AssertionError("M: (inv(W) - θ*S) is not PSD")              # the final solve! is not inside a try (:346)
    if st.value not in (nv.ST_OK, nv.ST_ITER_MAX):
        raise ArithmeticError(f"final iLEQG solve failed with status {st.value}")
    return th.value, x, l, nv.from_cm3(Lb, N, m, n), val.value
