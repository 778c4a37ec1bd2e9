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
        super().__init__(carrier, opts, max_batch=max(1, int(max_batch)), spec_eps=1, device=device)
        self.generic = problem
        self._opts = opts

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

    # ---- batched sweeps on host-built tiles (rat_dp_*_batch): the CE batch path of closure problems --------------------------------
    @staticmethod
    def _stack(aps):
        from . import _native as nv
        cat = np.concatenate
        return [cat([nv.f64(a.q_array) for a in aps]), cat([nv.f64(a.q_vec_array).ravel() for a in aps]), cat([nv.cm3(a.Q_array) for a in aps]),
                cat([nv.f64(a.r_array).ravel() for a in aps]), cat([nv.cm3(a.R_array) for a in aps]), cat([nv.cm3(a.P_array) for a in aps]),
                cat([nv.cm3(a.A_array) for a in aps]), cat([nv.cm3(a.B_array) for a in aps])]

    def _ensure_batch(self, B):
        """The carrier handle is created for one sample; a batch needs device buffers for B of them."""
        if B > self.max_batch:
            il.Context.__init__(self, self.problem, self._opts, max_batch=B, spec_eps=1, device=self.device)

    def dp_gain_sweep_batch(self, aps, theta, mu, delta):
        """solve_approximate_dp! (ileqg.jl:341-406) of B samples in ONE launch.  Returns status (B,), L (B, N, m, n), dl (B, N, m), mu, delta."""
        import ctypes as C
        from . import _native as nv
        B, n, m, N = len(aps), self.n, self.m, self.N
        self._ensure_batch(B)
        bufs = self._stack(aps)
        th, mu_c, de_c = nv.f64(theta).copy(), nv.f64(mu).copy(), nv.f64(delta).copy()
        Lb, dl, st = np.zeros(B * m * n * N), np.zeros((B, N, m)), np.zeros(B, np.int32)
        nv.check(nv.lib().rat_dp_gain_sweep_batch(self.h, C.c_int64(B), *[nv.P(b) for b in bufs], nv.P(th), nv.P(mu_c), nv.P(de_c), nv.P(Lb),
                                                  nv.P(dl), nv.PI(st)))
        L = np.stack([nv.from_cm3(Lb[b * m * n * N:(b + 1) * m * n * N], N, m, n) for b in range(B)])
        return st, L, dl, mu_c, de_c

    def dp_policy_eval_batch(self, aps, Ls, theta, mu):
        """solve_approximate_dp with dl = nothing (ileqg.jl:412-465) of B samples in ONE launch: value (Inf where M is not PD), status."""
        import ctypes as C
        from . import _native as nv
        B = len(aps)
        self._ensure_batch(B)
        bufs = self._stack(aps)
        Lc = np.concatenate([nv.cm3(L) for L in Ls])
        val, st = np.zeros(B), np.zeros(B, np.int32)
        nv.check(nv.lib().rat_dp_policy_eval_batch(self.h, C.c_int64(B), *[nv.P(b) for b in bufs], nv.P(Lc), nv.P(nv.f64(theta)), nv.P(nv.f64(mu)),
                                                   nv.P(val), nv.PI(st)))
        return val, st

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


def solve_closure_batch(problem: GenericRiskSensitiveProblem, x_0, u_array, theta_array, opts=None, ctx: GenericContext | None = None, **kw):
    """B complete solve!s (ileqg.jl:635-659) of a closure problem, one per theta -- what compute_cost fans out over workers
    (cross_entropy_bilevel_optimization.jl:144-192) -- with every Riccati sweep of the batch in ONE device launch.

    Per sample this is the reference's own sequence (initialize!, then step! = approximate_model -> solve_approximate_dp! -> line_search!
    until convergence); the samples advance in lockstep rounds only so that their sweeps can share launches: the host evaluates the
    closures (rollouts, linearisation) of every live sample, the device sweeps all of them at once.  Results equal B separate
    ileqg.solve_ calls.  Returns value (Inf where the reference would throw), status, iters, ls_evals."""
    from . import _native as nv
    th = nv.f64(theta_array)
    B, N, n, m = th.size, problem.N, problem.n, problem.m
    o = opts if opts is not None else il.make_opts(**kw)
    ctx = ctx or GenericContext(problem, o, max_batch=B)
    x0 = np.asarray(x_0, float)
    status, value = np.full(B, -1, np.int32), np.full(B, np.inf)
    iters, ls_evals = np.zeros(B, np.int32), np.zeros(B, np.int32)
    mu, delta, d_cur, eps_i = np.zeros(B), np.full(B, o.delta_0), np.full(B, np.inf), np.full(B, o.eps_init)
    # initialize!: one rollout / linearisation serves every sample (theta only enters the sweep)
    xs0 = ctx.rollout_open(x0, u_array)
    ap0 = ctx.approximate_model(np.asarray(u_array, float), xs0)
    x = [xs0.copy() for _ in range(B)]
    l = [np.array(u_array, float) for _ in range(B)]
    L = [np.zeros((N, m, n)) for _ in range(B)]
    v, st = ctx.dp_policy_eval_batch([ap0] * B, L, th, mu)
    status[st != 0] = nv.ST_M_NOT_PD_INIT
    value = np.where(st == 0, v, np.inf)
    dl = [None] * B
    eps, count, in_ls = eps_i.copy(), np.zeros(B, int), np.zeros(B, bool)
    while True:
        live = np.flatnonzero(status == -1)
        if live.size == 0:
            break
        need = [b for b in live if not in_ls[b]]
        if need:                                                                  # step!: approximate_model + solve_approximate_dp!
            iters[need] += 1
            aps = [ctx.approximate_model(l[b], x[b]) for b in need]
            stg, Lg, dlg, mug, deg = ctx.dp_gain_sweep_batch(aps, th[need], mu[need], delta[need])
            for i, b in enumerate(need):
                mu[b], delta[b] = mug[i], deg[i]
                if stg[i] != 0:
                    status[b], value[b] = stg[i], np.inf
                    continue
                L[b], dl[b] = Lg[i], dlg[i]
                eps[b], count[b], in_ls[b] = eps_i[b], 0, True
        cand = [b for b in np.flatnonzero(status == -1) if in_ls[b]]
        if not cand:
            continue
        trial = {}
        for b in cand:                                                            # line_search! candidate (ileqg.jl:504-521)
            count[b] += 1
            ls_evals[b] += 1
            xn, un = ctx.rollout_feedback(x[b], l[b] + eps[b] * dl[b], L[b])
            trial[b] = (xn, un, ctx.approximate_model(un, xn))
        vn, stn = ctx.dp_policy_eval_batch([trial[b][2] for b in cand], [L[b] for b in cand], th[cand], mu[cand])
        for i, b in enumerate(cand):
            if count[b] > 4000:
                status[b], value[b], in_ls[b] = nv.ST_LS_DIVERGED, np.inf, False
                continue
            if stn[i] != 0:
                eps[b] *= o.lam                                                   # :529-535 (no eps_min test, no history entry)
                continue
            new, cur = vn[i], value[b]
            if not (il._isapprox(new, cur) or new < cur):                         # :538
                eps[b] *= o.lam                                                   # :557
                if not eps[b] < o.eps_min:
                    continue
            xn, un, _ = trial[b]
            d_cur[b] = float(np.max(np.linalg.norm(l[b] - un, axis=1)))           # :539 / :559
            value[b], x[b], l[b], in_ls[b] = new, xn, un, False
            if o.adaptive_eps_init:                                               # :582-591
                if count[b] == 1:
                    eps_i[b] = min(o.eps_init, eps[b] / o.lam)
                else:
                    e = eps[b]
                    while e < o.eps_min:
                        e = e / o.lam
                    eps_i[b] = e
            if o.d > d_cur[b] and mu[b] <= o.mu_min:                              # :642
                status[b] = nv.ST_OK
            elif iters[b] == o.iter_max:                                          # :648
                status[b] = nv.ST_ITER_MAX
    value = np.where((status == nv.ST_OK) | (status == nv.ST_ITER_MAX), value, np.inf)
    return value, status, iters, ls_evals
