"""Generic-closure problems: ``FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N)`` with arbitrary host callables
(optimal_control_problems.jl:67-73), the fallback of SURVEY.md section 8f #3.

Closures cannot cross the C ABI into a kernel, so the split is the one the survey prescribes: the HOST runs the rollouts
(ileqg.jl:18-38, :62-87) and builds the ApproximationResult (ileqg.jl:258-322) -- from Jacobians the user's ``f`` returns
(``f_returns_jacobian``, ileqg.jl:24-31, :71-79, :302-311), from user-supplied cost derivatives, or by central differences where
neither is given (the reference uses ForwardDiff there, :265-273) -- and the DEVICE does what it is good at, the Riccati
sweeps, through the operator entry points ``rat_dp_gain_sweep`` / ``rat_dp_policy_eval`` (which take an ApproximationResult).
The solve loop is the reference's own initialize!/step!/line_search! sequence (``ileqg.solve_stepwise_``).

This is a correctness path, not a fast one: one trajectory per device call.  Problems that fit a compiled-in model family
(``LQRiskSensitiveProblem``, ``PowerLawRiskSensitiveProblem``) should use it -- then everything runs on the device."""
from __future__ import annotations

import numpy as np

from .problems import FiniteHorizonRiskSensitiveOptimalControlProblem, LQRiskSensitiveProblem
from . import ileqg as il


class GenericRiskSensitiveProblem(FiniteHorizonRiskSensitiveOptimalControlProblem):
    """f(x, u[, f_returns_jacobian]) -> x' [, A, B];  c(k, x, u) -> float;  h(x) -> float;  W(k) -> (n, n) SPD;  N steps.

    Optional exact derivatives (else central differences with step ``fd_step``):
      f_returns_jacobian  -- f(x, u, True) returns (x', A, B)                               (ileqg.jl:302-311)
      c_derivatives(k, x, u) -> (q_vec, Q, r_vec, R, P)   with P = d2c/du dx  (m x n)       (ileqg.jl:296-301)
      h_derivatives(x)       -> (q_vec, Q)                                                   (ileqg.jl:314-316)"""

    model = 0

    def __init__(self, f, c, h, W, N, n, m, f_returns_jacobian=False, c_derivatives=None, h_derivatives=None, fd_step=1e-5):
        self.f, self.c, self.h, self.W = f, c, h, W
        self.N, self.n, self.m = int(N), int(n), int(m)
        self.f_returns_jacobian = bool(f_returns_jacobian)
        self.c_derivatives, self.h_derivatives, self.fd_step = c_derivatives, h_derivatives, float(fd_step)
        self.Wtab = np.stack([np.asarray(W(k), dtype=np.float64) for k in range(self.N)])
        self.W_tv = bool(np.any(self.Wtab != self.Wtab[0]))

    # ---- derivatives ---------------------------------------------------------------------------------
    def _jac_f(self, x, u):
        if self.f_returns_jacobian:
            _, A, B = self.f(x, u, True)
            return np.asarray(A, float), np.asarray(B, float)
        hs, n, m = self.fd_step, self.n, self.m
        A, B = np.zeros((n, n)), np.zeros((n, m))
        for j in range(n):
            e = np.zeros(n); e[j] = hs
            A[:, j] = (np.asarray(self.f(x + e, u)) - np.asarray(self.f(x - e, u))) / (2 * hs)
        for j in range(m):
            e = np.zeros(m); e[j] = hs
            B[:, j] = (np.asarray(self.f(x, u + e)) - np.asarray(self.f(x, u - e))) / (2 * hs)
        return A, B

    @staticmethod
    def _grad_hess(fun, z, hs):
        d = z.size
        g, H = np.zeros(d), np.zeros((d, d))
        f0 = fun(z)
        E = np.eye(d) * hs
        fp = np.array([fun(z + E[i]) for i in range(d)])
        fm = np.array([fun(z - E[i]) for i in range(d)])
        g = (fp - fm) / (2 * hs)
        for i in range(d):
            H[i, i] = (fp[i] - 2 * f0 + fm[i]) / hs ** 2
            for j in range(i):
                H[i, j] = H[j, i] = (fun(z + E[i] + E[j]) - fun(z + E[i] - E[j]) - fun(z - E[i] + E[j]) + fun(z - E[i] - E[j])) / (4 * hs ** 2)
        return g, H

    def _cost_derivs(self, k, x, u):
        if self.c_derivatives is not None:
            return tuple(np.asarray(a, float) for a in self.c_derivatives(k, x, u))
        n = self.n
        g, H = self._grad_hess(lambda z: float(self.c(k, z[:n], z[n:])), np.concatenate([x, u]), self.fd_step * 10)
        return g[:n], H[:n, :n], g[n:], H[n:, n:], H[n:, :n]

    def _term_derivs(self, x):
        if self.h_derivatives is not None:
            return tuple(np.asarray(a, float) for a in self.h_derivatives(x))
        return self._grad_hess(lambda z: float(self.h(z)), np.asarray(x, float), self.fd_step * 10)


class GenericContext(il.Context):
    """Context whose rollouts and linearisation run the user's closures on the host; the Riccati sweeps run on the device through
    a carrier problem that only contributes the noise tables W(k)."""

    def __init__(self, problem: GenericRiskSensitiveProblem, opts=None, max_batch=1, spec_eps=1, device=0):
        n, m, N = problem.n, problem.m, problem.N
        carrier = LQRiskSensitiveProblem(np.zeros((n, n)), np.zeros((n, m)), Q=np.zeros((n, n)), R=np.eye(m), N=N,
                                         W=problem.Wtab if problem.W_tv else problem.Wtab[0])
        super().__init__(carrier, opts, max_batch=1, spec_eps=1, device=device)
        self.generic = problem

    def rollout_open(self, x0, u):                                   # simulate_dynamics  ileqg.jl:18-38
        p = self.generic
        x = np.zeros((p.N + 1, p.n))
        x[0] = x0
        for t in range(p.N):
            x[t + 1] = p.f(x[t], np.asarray(u[t], float))
        return x

    def rollout_feedback(self, xbar, l, L):                          # ileqg.jl:62-87
        p = self.generic
        xn, un = np.zeros((p.N + 1, p.n)), np.zeros((p.N, p.m))
        xn[0] = xbar[0]
        for t in range(p.N):
            un[t] = l[t] + L[t] @ (xn[t] - xbar[t])
            xn[t + 1] = p.f(xn[t], un[t])
        return xn, un

    def integrate_cost(self, x, u):                                  # ileqg.jl:115-124
        p = self.generic
        return float(sum(p.c(k, x[k], u[k]) for k in range(p.N)) + p.h(x[p.N]))

    def approximate_model(self, u, x):                               # ileqg.jl:258-322
        p = self.generic
        n, m, N = p.n, p.m, p.N
        ap = il.ApproximationResult(
            q_array=np.zeros(N + 1), q_vec_array=np.zeros((N + 1, n)), Q_array=np.zeros((N + 1, n, n)), r_array=np.zeros((N, m)),
            R_array=np.zeros((N, m, m)), P_array=np.zeros((N, m, n)), A_array=np.zeros((N, n, n)), B_array=np.zeros((N, n, m)),
            W_array=p.Wtab.copy())
        for k in range(N):
            xk, uk = np.asarray(x[k], float), np.asarray(u[k], float)
            ap.q_array[k] = p.c(k, xk, uk)
            ap.q_vec_array[k], ap.Q_array[k], ap.r_array[k], ap.R_array[k], ap.P_array[k] = p._cost_derivs(k, xk, uk)
            ap.A_array[k], ap.B_array[k] = p._jac_f(xk, uk)
        ap.q_array[N] = p.h(np.asarray(x[N], float))
        ap.q_vec_array[N], ap.Q_array[N] = p._term_derivs(np.asarray(x[N], float))
        return ap

    def solve(self, x0, u, theta, hist_cap=4096):
        raise NotImplementedError("generic closures are solved by ileqg.solve_ (host-driven initialize!/step! loop)")

    # The remaining device-family entry points of Context would run on the CARRIER (A = 0, B = 0, R = I), not on the user's f, c, h:
    # they fail loudly instead.
    def _carrier_only(self, name):
        raise NotImplementedError(f"{name} runs a compiled-in model family on the device; a generic-closure problem has none "
                                  "(use ileqg.solve_ / solve_closure_batch, or a LQ / power-law family)")

    def rollout_noisy(self, *a, **k):
        self._carrier_only("rollout_noisy (simulate_dynamics with rng)")

    def solve_batch(self, *a, **k):
        self._carrier_only("solve_batch")

    def solve_batch_dev(self, *a, **k):
        self._carrier_only("solve_batch_dev")

    def compute_cost_dev(self, *a, **k):
        self._carrier_only("compute_cost_dev")

    def compute_cost_enqueue(self, *a, **k):
        self._carrier_only("compute_cost_enqueue")

    def set_initial(self, *a, **k):
        self._carrier_only("set_initial")
