"""Optimal-control problem definitions -- host-side mirror of the reference's
``src/optimal_control_problems.jl``.

The reference's ``FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N)``
(optimal_control_problems.jl:67-73) is a struct of arbitrary Julia closures that ForwardDiff
differentiates (ileqg.jl:265-273).  Closures cannot cross a C ABI into a GPU kernel, so the
MI355X path compiles in *model families* whose derivatives are analytic on the device:

* ``LQRiskSensitiveProblem``       f = A x + B u + kappa x.^3,  quadratic time-varying c_k, quadratic h
* ``PowerLawRiskSensitiveProblem`` f = x.^a + u.^b, c = cx sum(x.^p) + cu sum(u.^pu), h = const
                                   (the nonlinear system of test/ileqg_test.jl:151-155)

Both still expose the reference's field names ``f, c, h, W, N`` as Python callables, so code
written against the reference struct keeps working on the host.
"""
from __future__ import annotations

import numpy as np

MODEL_LQ = 1
MODEL_POWERLAW = 2


class OptimalControlProblem:
    """Abstract base type (optimal_control_problems.jl:12)."""


class FiniteHorizonRiskSensitiveOptimalControlProblem(OptimalControlProblem):
    """Field-compatible base of the device model families (optimal_control_problems.jl:67-73).

    Sub-classes fill ``f(x, u, f_returns_jacobian=False)``, ``c(k, x, u)``, ``h(x)``, ``W(k)``, ``N``
    plus the flat parameter tables the C ABI consumes (``c_tables()``).
    """

    model = 0
    n = 0
    m = 0
    N = 0

    def c_tables(self) -> dict:
        raise NotImplementedError(
            "generic closures cannot be sent to the GPU; use LQRiskSensitiveProblem or "
            "PowerLawRiskSensitiveProblem (SURVEY.md section 7, 'hard parts')"
        )


def _colmajor(a: np.ndarray) -> np.ndarray:
    """Flat column-major (Julia-native) buffer, time slowest for 3-D stacks."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim <= 1:
        return np.ascontiguousarray(a).ravel()
    if a.ndim == 2:
        return np.ascontiguousarray(a.T).ravel()
    return np.ascontiguousarray(a.transpose(0, 2, 1)).ravel()


def _timemajor(a: np.ndarray) -> np.ndarray:
    """Flat buffer of a (possibly time-varying) VECTOR table: a Vector{Vector} of Julia, time slowest -- entry (k, i) at k * len + i.
    (A 2-D (N, len) table is NOT a matrix: _colmajor would transpose it.)"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).ravel()


class LQRiskSensitiveProblem(FiniteHorizonRiskSensitiveOptimalControlProblem):
    r"""x' = A x + B u + kappa x.^3 + w,  w ~ N(0, W(k)).

    c(k,x,u) = 1/2 x'Q_k x + 1/2 u'R_k u + u'P_k x + qv_k'x + rv_k'u + q0_k,
    h(x)     = 1/2 x'Qf x + qvf'x + q0f.
    Cost tables may be constant (shape (n,n) ...) or time-varying (leading axis N).
    """

    model = MODEL_LQ

    def __init__(self, A, B, Q, R, N, W, P=None, qv=None, rv=None, q0=None, Qf=None, qvf=None, q0f=0.0,
                 kappa=0.0):
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        self.n, self.m, self.N = A.shape[0], B.shape[1], int(N)
        n, m = self.n, self.m
        assert A.shape == (n, n) and B.shape == (n, m) and self.N > 0
        Q = np.asarray(Q, dtype=np.float64)
        self.cost_tv = Q.ndim == 3
        lead = (self.N,) if self.cost_tv else ()

        def tab(v, shape):
            if v is None:
                return np.zeros(lead + shape)
            v = np.asarray(v, dtype=np.float64)
            if v.shape == shape and self.cost_tv:
                v = np.broadcast_to(v, lead + shape).copy()
            assert v.shape == lead + shape, (v.shape, lead + shape)
            return v

        self.A, self.B = A, B
        self.Q, self.R = tab(Q, (n, n)), tab(R, (m, m))
        self.P, self.qv, self.rv = tab(P, (m, n)), tab(qv, (n,)), tab(rv, (m,))
        self.q0 = tab(q0, ())
        self.Qf = np.zeros((n, n)) if Qf is None else np.asarray(Qf, dtype=np.float64)
        self.qvf = np.zeros(n) if qvf is None else np.asarray(qvf, dtype=np.float64)
        self.q0f = float(q0f)
        self.kappa = float(kappa)
        W = np.asarray(W, dtype=np.float64)
        self.W_tv = W.ndim == 3
        assert W.shape == ((self.N, n, n) if self.W_tv else (n, n))
        self.Wtab = W

    # ---- the reference's closure fields -------------------------------------------------
    def f(self, x, u, f_returns_jacobian=False):
        x, u = np.asarray(x, float), np.asarray(u, float)
        xn = self.A @ x + self.B @ u + self.kappa * x ** 3
        if f_returns_jacobian:
            return xn, self.A + np.diag(3.0 * self.kappa * x ** 2), self.B.copy()
        return xn

    def _k(self, tab, k):
        return tab[k] if self.cost_tv else tab

    def c(self, k, x, u):
        x, u = np.asarray(x, float), np.asarray(u, float)
        return float(0.5 * x @ self._k(self.Q, k) @ x + 0.5 * u @ self._k(self.R, k) @ u
                     + u @ self._k(self.P, k) @ x + self._k(self.qv, k) @ x + self._k(self.rv, k) @ u
                     + self._k(self.q0, k))

    def h(self, x):
        x = np.asarray(x, float)
        return float(0.5 * x @ self.Qf @ x + self.qvf @ x + self.q0f)

    def W(self, k):
        return self.Wtab[k] if self.W_tv else self.Wtab

    def c_tables(self) -> dict:
        return dict(model=MODEL_LQ, n=self.n, m=self.m, N=self.N, cost_tv=int(self.cost_tv), W_tv=int(self.W_tv),
                    A=_colmajor(self.A), B=_colmajor(self.B), Q=_colmajor(self.Q), R=_colmajor(self.R),
                    P=_colmajor(self.P), qv=_timemajor(self.qv), rv=_timemajor(self.rv),
                    q0=_colmajor(np.atleast_1d(self.q0)), Qf=_colmajor(self.Qf), qvf=_colmajor(self.qvf),
                    q0f=self.q0f, kappa=self.kappa, W=_colmajor(self.Wtab),
                    pl_a=0.0, pl_b=0.0, pl_p=0.0, pl_pu=0.0, pl_cx=0.0, pl_cu=0.0, pl_h=0.0)


class PowerLawRiskSensitiveProblem(FiniteHorizonRiskSensitiveOptimalControlProblem):
    """f = x.^a + u.^b (n == m), c = cx*sum(x.^p) + cu*sum(u.^pu), h = hconst  (test/ileqg_test.jl:151-155)."""

    model = MODEL_POWERLAW

    def __init__(self, n, N, W, a=1.3, b=1.5, p=2.5, pu=None, cx=1.0, cu=1.0, hconst=1.0):
        self.n = self.m = int(n)
        self.N = int(N)
        self.a, self.b, self.p = float(a), float(b), float(p)
        self.pu = float(p if pu is None else pu)
        self.cx, self.cu, self.hconst = float(cx), float(cu), float(hconst)
        W = np.asarray(W, dtype=np.float64)
        self.W_tv = W.ndim == 3
        assert W.shape == ((self.N, self.n, self.n) if self.W_tv else (self.n, self.n))
        self.Wtab = W

    def f(self, x, u, f_returns_jacobian=False):
        x, u = np.asarray(x, float), np.asarray(u, float)
        xn = x ** self.a + u ** self.b
        if f_returns_jacobian:
            return xn, np.diag(self.a * x ** (self.a - 1)), np.diag(self.b * u ** (self.b - 1))
        return xn

    def c(self, k, x, u):
        x, u = np.asarray(x, float), np.asarray(u, float)
        return float(self.cx * np.sum(x ** self.p) + self.cu * np.sum(u ** self.pu))

    def h(self, x):
        return self.hconst

    def W(self, k):
        return self.Wtab[k] if self.W_tv else self.Wtab

    def c_tables(self) -> dict:
        n, m = self.n, self.m
        z = np.zeros
        return dict(model=MODEL_POWERLAW, n=n, m=m, N=self.N, cost_tv=0, W_tv=int(self.W_tv),
                    A=z(n * n), B=z(n * m), Q=z(n * n), R=z(m * m), P=z(m * n), qv=z(n), rv=z(m), q0=z(1),
                    Qf=z(n * n), qvf=z(n), q0f=0.0, kappa=0.0, W=_colmajor(self.Wtab),
                    pl_a=self.a, pl_b=self.b, pl_p=self.p, pl_pu=self.pu, pl_cx=self.cx, pl_cu=self.cu,
                    pl_h=self.hconst)


def synthetic_lq_problem(n=12, m=4, N=50, rho=0.9, w=1e-3, r_weight=0.1, seed=0, kappa=0.0):
    """The seeded 'LQ-plus-noise' benchmark problem of SURVEY.md section 8(d).

    A = rho * Qorth (normal matrix), B = randn/sqrt(n), c = 1/2 x'x + 1/2 r_weight u'u, h = 1/2 x'x,
    W = w I, x0 = randn(n), u = 0.  Generator: numpy default_rng(seed) (PCG64), documented in DESIGN.md.
    Returns (problem, x0, u_array).
    """
    rng = np.random.default_rng(seed)
    Qo, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = rho * Qo
    B = rng.standard_normal((n, m)) / np.sqrt(n)
    x0 = rng.standard_normal(n)
    prob = LQRiskSensitiveProblem(A, B, Q=np.eye(n), R=r_weight * np.eye(m), N=N, W=w * np.eye(n), Qf=np.eye(n),
                                  kappa=kappa)
    return prob, x0, np.zeros((N, m))


class FiniteHorizonGenerativeOptimalControlProblem(OptimalControlProblem):
    """Field-compatible base of the generative model family (optimal_control_problems.jl:126-131):
    ``f_stochastic(x, u, rng, use_true_model=False)``, ``c(k, x, u)``, ``h(x)``, ``N``."""


class LQGenerativeProblem(FiniteHorizonGenerativeOptimalControlProblem):
    r"""f_stochastic(x, u, rng, use_true_model) = A x + B u + kappa x.^3 + w  (device family of the PETS path).

    noise = ("gaussian", mean, cov) or ("uniform", lo, hi)  (the latter is ``x + u + rand(rng, n)`` of test/pets_test.jl:15);
    true_noise = (w2, mean2, cov2): with ``use_true_model`` the noise is N(mean2, cov2) with probability w2, else the model
    noise (2-component mixture of the docs example).  c(k,x,u) = LQ quadratic form + l1u*sum(abs.(u)); h quadratic."""

    def __init__(self, A, B, N, noise, Q=None, R=None, P=None, qv=None, rv=None, q0=None, Qf=None, qvf=None, q0f=0.0,
                 kappa=0.0, l1u=0.0, true_noise=None):
        A = np.asarray(A, float)
        n, m = A.shape[0], np.asarray(B).shape[1]
        self.lq = LQRiskSensitiveProblem(A, B, Q=np.zeros((n, n)) if Q is None else Q, R=np.zeros((m, m)) if R is None else R,
                                         N=N, W=np.eye(n), P=P, qv=qv, rv=rv, q0=q0, Qf=Qf, qvf=qvf, q0f=q0f, kappa=kappa)
        self.n, self.m, self.N = n, m, int(N)
        self.l1u = float(l1u)
        self.noise = noise
        self.true_noise = true_noise
        if noise[0] == "gaussian":
            self.noise_kind, self.nmean = 0, np.asarray(noise[1], float)
            self.nchol = np.linalg.cholesky(np.asarray(noise[2], float))
            self.nlo = self.nhi = 0.0
        elif noise[0] == "uniform":
            self.noise_kind, self.nmean, self.nchol = 1, np.zeros(n), np.zeros((n, n))
            self.nlo, self.nhi = float(noise[1]), float(noise[2])
        else:
            raise ValueError(noise[0])
        if true_noise is not None:
            if self.noise_kind != 0 and float(true_noise[0]) > 0:
                raise ValueError("the true-model mixture is defined over Gaussian model noise")
            self.tw2, self.tmean2 = float(true_noise[0]), np.asarray(true_noise[1], float)
            self.tchol2 = np.linalg.cholesky(np.asarray(true_noise[2], float))
        else:
            self.tw2, self.tmean2, self.tchol2 = 0.0, np.zeros(n), np.zeros((n, n))

    def f_stochastic(self, x, u, rng, use_true_model=False):
        x = np.asarray(x, float)
        xn = self.lq.f(x, u)
        if use_true_model and self.tw2 > 0 and rng.random() < self.tw2:
            return xn + self.tmean2 + self.tchol2 @ rng.standard_normal(self.n)
        if self.noise_kind == 0:
            return xn + self.nmean + self.nchol @ rng.standard_normal(self.n)
        return xn + self.nlo + (self.nhi - self.nlo) * rng.random(self.n)

    def c(self, k, x, u):
        return self.lq.c(k, x, u) + self.l1u * float(np.sum(np.abs(u)))

    def h(self, x):
        return self.lq.h(x)

    def gen_tables(self) -> dict:
        t = dict(self.lq.c_tables())
        t.update(l1u=self.l1u, noise_kind=self.noise_kind, nmean=_colmajor(self.nmean), nchol=_colmajor(self.nchol),
                 nlo=self.nlo, nhi=self.nhi, tw2=self.tw2, tmean2=_colmajor(self.tmean2), tchol2=_colmajor(self.tchol2))
        return t
