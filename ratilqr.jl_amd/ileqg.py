"""Host-side mirror of the reference's ``src/ileqg.jl`` API over the C ABI (include/ratilqr.h).

Same names as the reference's exports (src/RATiLQR.jl:20-53); Julia's ``f!`` becomes ``f_``.
Greek keyword arguments are spelled out (``mu_min`` for μ_min, ``Delta_0`` for Δ_0, ``lam`` for λ,
``eps_init`` for ϵ_init ...).  Array conventions on the Python side: a ``Vector{Vector}`` is a 2-D
array ``[t, i]``, a ``Vector{Matrix}`` a 3-D array ``[t, row, col]``.

All numerics run in libratilqr_hip.so on the GPU.  ``solve_`` / batched solves use the fused device
state machine; ``initialize_`` / ``step_`` / ``line_search_`` compose the operator entry points exactly
as the reference composes its functions, so the unit tests of test/ileqg_test.jl can be restated 1:1.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass

import numpy as np

from . import _native as nv
from .problems import FiniteHorizonRiskSensitiveOptimalControlProblem

SQRT_EPS = 1.4901161193847656e-8


class Context:
    """One rat_handle bound to one problem (device buffers sized for max_batch samples x spec_eps step sizes)."""

    def __init__(self, problem, opts: nv.IleqgOpts | None = None, max_batch=1, spec_eps=1, device=0):
        L = nv.lib()
        self.problem = problem
        self.n, self.m, self.N = problem.n, problem.m, problem.N
        self.max_batch, self.spec_eps, self.device = int(max_batch), int(spec_eps), int(device)
        self.h = C.c_void_p()
        nv.check(L.rat_create(C.byref(opts) if opts is not None else None, self.max_batch, self.spec_eps,
                              self.device, C.byref(self.h)))
        self._fin = weakref.finalize(self, L.rat_destroy, self.h)
        desc, self._keep = nv.make_desc(problem)
        nv.check(L.rat_problem_set(self.h, C.byref(desc)))

    def set_opts(self, opts):
        nv.check(nv.lib().rat_set_ileqg_opts(self.h, C.byref(opts)))

    def set_problem(self, problem):
        """Re-bind the handle to another problem of the same model families (rat_problem_set on the live handle: device buffers are
        kept when n, m, N are unchanged -- the receding-horizon pattern of re-setting the tables every control step)."""
        desc, keep = nv.make_desc(problem)
        nv.check(nv.lib().rat_problem_set(self.h, C.byref(desc)))
        self.problem, self._keep = problem, keep
        self.n, self.m, self.N = problem.n, problem.m, problem.N

    # ---- operator forms --------------------------------------------------------------------------
    def rollout_open(self, x0, u):
        x = np.zeros((self.N + 1, self.n))
        dom = C.c_int32()
        nv.check(nv.lib().rat_rollout_open(self.h, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), nv.P(x), C.byref(dom)))
        if dom.value:
            raise ArithmeticError("DomainError in simulate_dynamics")
        return x

    def rollout_feedback(self, xbar, l, L):
        xn, un = np.zeros((self.N + 1, self.n)), np.zeros((self.N, self.m))
        dom = C.c_int32()
        nv.check(nv.lib().rat_rollout_feedback(self.h, nv.P(nv.f64(xbar)), nv.P(nv.f64(l)), nv.P(nv.cm3(L)),
                                               nv.P(xn), nv.P(un), C.byref(dom)))
        if dom.value:
            raise ArithmeticError("DomainError in simulate_dynamics")
        return xn, un

    def rollout_noisy(self, x_nom, l, L=None, K=None, z=None, seed=0, want_x=True, want_u=True):
        """K Monte-Carlo rollouts under process noise w ~ N(0, W(k)) (ileqg.jl:44-55 open loop with L=None and x_nom = x_0,
        :94-109 under the affine policy).  z: injected N(0,1) draws of shape (K, N, n), or None for the device generator.
        Returns x (K, N+1, n), u (K, N, m), cost (K,) -- x/u are None when not wanted."""
        if z is not None:
            z = nv.f64(z)
            K = z.shape[0]
        K = int(K)
        x = np.zeros((K, self.N + 1, self.n)) if want_x else None
        u = np.zeros((K, self.N, self.m)) if want_u else None
        cost = np.zeros(K)
        dom = C.c_int32()
        nv.check(nv.lib().rat_rollout_noisy(self.h, nv.P(nv.f64(x_nom)), nv.P(nv.f64(l)), nv.P(nv.cm3(L)) if L is not None else None,
                                            C.c_int64(K), nv.P(z), C.c_uint64(int(seed)), nv.P(x), nv.P(u), nv.P(cost), C.byref(dom)))
        return x, u, cost, bool(dom.value)

    def integrate_cost(self, x, u):
        out = C.c_double()
        nv.check(nv.lib().rat_integrate_cost(self.h, nv.P(nv.f64(x)), nv.P(nv.f64(u)), C.byref(out)))
        return out.value

    def approximate_model(self, u, x):
        n, m, N = self.n, self.m, self.N
        b = dict(q=np.zeros(N + 1), qv=np.zeros(n * (N + 1)), Q=np.zeros(n * n * (N + 1)), r=np.zeros(m * N),
                 R=np.zeros(m * m * N), P=np.zeros(m * n * N), A=np.zeros(n * n * N), B=np.zeros(n * m * N),
                 W=np.zeros(n * n * N))
        dom = C.c_int32()
        nv.check(nv.lib().rat_approximate_model(self.h, nv.P(nv.f64(u)), nv.P(nv.f64(x)), *[nv.P(b[k]) for k in
                                                ("q", "qv", "Q", "r", "R", "P", "A", "B", "W")], C.byref(dom)))
        if dom.value:
            raise ArithmeticError("DomainError in approximate_model")
        return ApproximationResult(
            q_array=b["q"], q_vec_array=b["qv"].reshape(N + 1, n), Q_array=nv.from_cm3(b["Q"], N + 1, n, n),
            r_array=b["r"].reshape(N, m), R_array=nv.from_cm3(b["R"], N, m, m), P_array=nv.from_cm3(b["P"], N, m, n),
            A_array=nv.from_cm3(b["A"], N, n, n), B_array=nv.from_cm3(b["B"], N, n, m),
            W_array=nv.from_cm3(b["W"], N, n, n))

    def _approx_ptrs(self, ap):
        bufs = [nv.f64(ap.q_array), nv.f64(ap.q_vec_array), nv.cm3(ap.Q_array), nv.f64(ap.r_array), nv.cm3(ap.R_array),
                nv.cm3(ap.P_array), nv.cm3(ap.A_array), nv.cm3(ap.B_array)]
        return bufs, [nv.P(b) for b in bufs]

    def _dp_out(self):
        n, m, N = self.n, self.m, self.N
        return dict(s=np.zeros(N + 1), sv=np.zeros(n * (N + 1)), S=np.zeros(n * n * (N + 1)), g=np.zeros(m * N),
                    G=np.zeros(m * n * N), H=np.zeros(m * m * N))

    def _dp_result(self, o):
        n, m, N = self.n, self.m, self.N
        return DynamicProgrammingResult(
            s_array=o["s"], s_vec_array=o["sv"].reshape(N + 1, n), S_array=nv.from_cm3(o["S"], N + 1, n, n),
            g_array=o["g"].reshape(N, m), G_array=nv.from_cm3(o["G"], N, m, n), H_array=nv.from_cm3(o["H"], N, m, m))

    def dp_gain_sweep(self, ap, theta, mu, delta):
        n, m, N = self.n, self.m, self.N
        keep, ptrs = self._approx_ptrs(ap)
        mu_c, de_c, st = C.c_double(mu), C.c_double(delta), C.c_int32()
        Lb, dl, o = np.zeros(m * n * N), np.zeros((N, m)), self._dp_out()
        nv.check(nv.lib().rat_dp_gain_sweep(self.h, *ptrs, C.c_double(theta), C.byref(mu_c), C.byref(de_c), nv.P(Lb),
                                            nv.P(dl), C.byref(st), *[nv.P(o[k]) for k in ("s", "sv", "S", "g", "G", "H")]))
        return st.value, nv.from_cm3(Lb, N, m, n), dl, self._dp_result(o), mu_c.value, de_c.value

    def dp_policy_eval(self, ap, L, dl, theta, mu):
        keep, ptrs = self._approx_ptrs(ap)
        st, o = C.c_int32(), self._dp_out()
        Lc = nv.cm3(L)
        dlc = None if dl is None else nv.f64(dl)
        nv.check(nv.lib().rat_dp_policy_eval(self.h, *ptrs, nv.P(Lc), nv.P(dlc), C.c_double(theta), C.c_double(mu),
                                             C.byref(st), *[nv.P(o[k]) for k in ("s", "sv", "S", "g", "G", "H")]))
        return st.value, self._dp_result(o)

    # ---- fused solves ----------------------------------------------------------------------------
    def solve(self, x0, u, theta, hist_cap=4096):
        n, m, N = self.n, self.m, self.N
        x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
        val, st, it, hn = C.c_double(), C.c_int32(), C.c_int32(), C.c_int64()
        while True:           # eps_history is unbounded in the reference (ileqg.jl:537): when it did not fit, grow and re-run (deterministic)
            hist = np.zeros((hist_cap, 2))
            nv.check(nv.lib().rat_ileqg_solve(self.h, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), C.c_double(theta), nv.P(x), nv.P(l),
                                              nv.P(Lb), C.byref(val), C.byref(st), C.byref(it), nv.P(hist),
                                              C.c_int64(hist_cap), C.byref(hn)))
            if hn.value <= hist_cap:
                break
            hist_cap = int(hn.value)
        return dict(x=x, l=l, L=nv.from_cm3(Lb, N, m, n), value=val.value, status=st.value, iters=it.value,
                    eps_history=hist[: min(hn.value, hist_cap)].copy(), hist_n=hn.value)

    def solve_batch(self, x0, u, theta):
        theta = nv.f64(theta)
        B = theta.size
        value, status = np.zeros(B), np.zeros(B, np.int32)
        iters, ls = np.zeros(B, np.int32), np.zeros(B, np.int32)
        nv.check(nv.lib().rat_ileqg_solve_batch(self.h, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), nv.P(theta), C.c_int64(B),
                                                nv.P(value), nv.PI(status), nv.PI(iters), nv.PI(ls)))
        return value, status, iters, ls

    def set_initial(self, x0, u):
        nv.check(nv.lib().rat_set_initial(self.h, nv.P(nv.f64(x0)), nv.P(nv.f64(u))))

    def solve_batch_dev(self, theta_ptr, B, value_ptr, status_ptr=None, iters_ptr=None, ls_ptr=None):
        """Device-pointer form (ints): inputs/outputs stay in HBM."""
        nv.check(nv.lib().rat_ileqg_solve_batch_dev(self.h, C.c_void_p(theta_ptr), C.c_int64(B), C.c_void_p(value_ptr),
                                                    C.c_void_p(status_ptr), C.c_void_p(iters_ptr), C.c_void_p(ls_ptr)))

    def compute_cost_dev(self, theta_ptr, B, kl_bound, cost_ptr):
        """compute_cost (cross_entropy...jl:173-195), device-pointer form: cost = value + kl_bound / theta stays in HBM."""
        nv.check(nv.lib().rat_ce_compute_cost_dev(self.h, C.c_void_p(theta_ptr), C.c_int64(B), C.c_double(kl_bound),
                                                  C.c_void_p(cost_ptr)))

    def compute_cost_enqueue(self, theta_ptr, B, kl_bound, cost_ptr):
        """Stream-ordered compute_cost_dev: returns once the batch is enqueued on ``self.stream`` (hipStream_t as int)."""
        nv.check(nv.lib().rat_ce_compute_cost_enqueue(self.h, C.c_void_p(theta_ptr), C.c_int64(B), C.c_double(kl_bound),
                                                      C.c_void_p(cost_ptr)))

    def compute_cost_enqueue_ex(self, theta_ptr, B, kl_bound, cost_ptr, status_ptr=None, iters_ptr=None, ls_ptr=None):
        """compute_cost_enqueue with the per-sample status / iteration / line-search counts written beside the costs."""
        nv.check(nv.lib().rat_ce_compute_cost_enqueue_ex(self.h, C.c_void_p(theta_ptr), C.c_int64(B), C.c_double(kl_bound), C.c_void_p(cost_ptr),
                                                         C.c_void_p(status_ptr), C.c_void_p(iters_ptr), C.c_void_p(ls_ptr)))

    # ---- execution path (include/ratilqr.h RAT_PATH_*; results are identical on all of them) -------
    PATHS = {"auto": 0, "rounds": 1, "fused": 2, "block": 3}

    def set_path(self, path):
        """Fix the execution path of this handle's batched solves: "auto" | "rounds" | "fused" | "block" (rat_set_path).  Moving between
        the single-launch E = 1 kernels and the round-based path re-lays the state: the initial trajectory must be given again."""
        nv.check(nv.lib().rat_set_path(self.h, C.c_int32(self.PATHS[path] if isinstance(path, str) else int(path))))

    def debug_set(self, key, value):
        """An execution switch of the handle (rat_debug_set; keys in include/ratilqr.h): tests, A/B tools, bench.py's contract leg."""
        nv.check(nv.lib().rat_debug_set(self.h, key.encode(), C.c_int64(int(value))))

    def debug_get(self, key):
        v = C.c_int64(0)
        nv.check(nv.lib().rat_debug_get(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def get_path(self, B):
        """Which path a batch of B samples takes: "rounds" | "fused" | "block" | "wide"."""
        r = int(nv.lib().rat_get_path(self.h, C.c_int64(int(B))))
        return {1: "rounds", 2: "fused", 3: "block", 4: "wide"}.get(r)

    @property
    def stream(self):
        """The handle's HIP stream (hipStream_t) as an integer, e.g. for ``torch.cuda.ExternalStream``."""
        return int(nv.lib().rat_stream(self.h) or 0)

    # ---- measurement -----------------------------------------------------------------------------
    def profile(self, on=True, kinds=None):
        """HIP-event timing of kernel launches; ``kinds`` (names from _native.K_NAMES) restricts what is recorded."""
        flag = int(bool(on))
        if on and kinds is not None:
            mask = 0
            for k in kinds:
                mask |= 1 << nv.K_NAMES.index(k)
            flag = (mask << 1) | 1
        nv.check(nv.lib().rat_profile_enable(self.h, flag))

    def profile_reset(self):
        nv.check(nv.lib().rat_profile_reset(self.h))

    def profile_get(self):
        nk = len(nv.K_NAMES)
        la = (C.c_int64 * nk)(); tr = (C.c_int64 * nk)(); ms = (C.c_double * nk)()
        nv.check(nv.lib().rat_profile_get(self.h, la, tr, ms))
        return {nv.K_NAMES[k]: dict(launches=la[k], trajectories=tr[k], ms=ms[k]) for k in range(nk)}

    def layout_info(self):
        v = [C.c_int64() for _ in range(4)]
        nv.check(nv.lib().rat_layout_info(self.h, *[C.byref(x) for x in v]))
        return dict(tile_bytes=v[0].value, L_bytes=v[1].value, x_bytes=v[2].value, u_bytes=v[3].value)


@dataclass
class ApproximationResult:                 # ileqg.jl:242-252
    q_array: np.ndarray
    q_vec_array: np.ndarray
    Q_array: np.ndarray
    r_array: np.ndarray
    R_array: np.ndarray
    P_array: np.ndarray
    A_array: np.ndarray
    B_array: np.ndarray
    W_array: np.ndarray


@dataclass
class DynamicProgrammingResult:            # ileqg.jl:328-335
    s_array: np.ndarray
    s_vec_array: np.ndarray
    S_array: np.ndarray
    g_array: np.ndarray
    G_array: np.ndarray
    H_array: np.ndarray


_ctx_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _ctx(problem) -> Context:
    """Default context of a problem for the stateless reference functions (simulate_dynamics, ...)."""
    c = _ctx_cache.get(problem)
    if c is None:
        c = make_context(problem)
        _ctx_cache[problem] = c
    return c


def make_context(problem, opts=None, max_batch=1, spec_eps=1, device=0) -> Context:
    """Context of a device model family, or the host-closure context of a generic problem (generic.py)."""
    if getattr(problem, "model", 0) == 0:
        from .generic import GenericContext
        return GenericContext(problem, opts, max_batch=max_batch, spec_eps=spec_eps, device=device)
    return Context(problem, opts, max_batch=max_batch, spec_eps=spec_eps, device=device)


def make_opts(mu_min=1e-6, Delta_0=2.0, lam=0.5, d=1e-2, iter_max=100, eps_init=1.0, adaptive_eps_init=False,
              eps_min=1e-6) -> nv.IleqgOpts:
    # the @assert block of ileqg.jl:195-201
    assert 0 < lam < 1, "λ has to be in (0, 1)"
    assert d > 0, "d > 0 is necessary"
    assert mu_min > 0, "μ_min > 0 is necessary"
    assert Delta_0 > 0, "Δ_0 > 0 is necessary"
    assert 0 < eps_init <= 1, "ϵ_init has to be in (0, 1]"
    assert eps_init > eps_min, "ϵ_init > ϵ_min is necessary"
    assert 0 < eps_min < 1, "ϵ_min has to be in (0, 1)"
    o = nv.IleqgOpts()
    o.mu_min, o.delta_0, o.lam, o.d, o.iter_max = mu_min, Delta_0, lam, d, int(iter_max)
    o.eps_init, o.eps_min, o.adaptive_eps_init = eps_init, eps_min, int(bool(adaptive_eps_init))
    return o


class ILEQGSolver:
    """ILEQGSolver(problem; kwargs...)  (ileqg.jl:164-208)."""

    def __init__(self, problem: FiniteHorizonRiskSensitiveOptimalControlProblem, mu_min=1e-6, Delta_0=2.0, lam=0.5,
                 d=1e-2, iter_max=100, eps_init=1.0, adaptive_eps_init=False, eps_min=1e-6, f_returns_jacobian=False,
                 max_batch=1, spec_eps=1, device=0):
        self.opts = make_opts(mu_min, Delta_0, lam, d, iter_max, eps_init, adaptive_eps_init, eps_min)
        self.mu_min, self.mu, self.Delta_0, self.Delta = mu_min, mu_min, Delta_0, Delta_0      # :206
        self.lam, self.d, self.iter_max = lam, d, int(iter_max)
        self.eps_init_auto, self.eps_init, self.eps_min = bool(adaptive_eps_init), eps_init, eps_min
        self.f_returns_jacobian = f_returns_jacobian    # analytic Jacobians are always used on the device
        self.x_array = self.l_array = self.L_array = None
        self.A_array = self.B_array = None
        self.value_current, self.iter_current, self.d_current = np.inf, 0, np.inf
        self.eps_history = []
        self.eps_init_init = eps_init
        self.status = None
        self._ctx_args = dict(max_batch=max_batch, spec_eps=spec_eps, device=device)
        self.ctx = make_context(problem, self.opts, **self._ctx_args)

    def context(self, problem) -> "Context":
        """The device context of `problem`.  The reference's solve!/initialize!/step!/line_search! take every table from their
        `problem` argument (ileqg.jl:214, 494, 598, 635), so a solver built on one problem and called with another must follow the
        argument: the context (device tables) is rebuilt when the problem object differs from the one it was made for."""
        bound = getattr(self.ctx, "generic", None) or self.ctx.problem
        if problem is not bound:
            self.ctx = make_context(problem, self.opts, **self._ctx_args)
        return self.ctx


# ---- the reference's free functions ----------------------------------------------------------------
def simulate_dynamics(problem, a, b, c=None, f_returns_jacobian=False):
    """simulate_dynamics(problem, x_0, u_array) (ileqg.jl:18-38) or
    simulate_dynamics(problem, x_array, l_array, L_array) (ileqg.jl:62-87)."""
    ctx = _ctx(problem)
    if c is None:
        return ctx.rollout_open(a, b)
    return ctx.rollout_feedback(a, b, c)


def simulate_dynamics_noisy(problem, a, b, c=None, K=1, z=None, seed=0):
    """The rng methods of simulate_dynamics, K rollouts at once: simulate_dynamics(problem, x_0, u_array, rng) (ileqg.jl:44-55)
    or simulate_dynamics(problem, x_array, l_array, L_array, rng) (ileqg.jl:94-109).  The rng is an injected standard-normal
    array z of shape (K, N, n) or a seed of the device generator.  Returns (x, cost) or (x, u, cost), rollout index first."""
    ctx = _ctx(problem)
    x, u, cost, dom = ctx.rollout_noisy(a, b, c, K=K, z=z, seed=seed)
    if dom:
        raise ArithmeticError("DomainError in simulate_dynamics")
    return (x, cost) if c is None else (x, u, cost)


def integrate_cost(problem, x_array, u_array):          # ileqg.jl:115-124
    return _ctx(problem).integrate_cost(x_array, u_array)


def approximate_model(problem, u_array, x_array, A_array_input=None, B_array_input=None):   # ileqg.jl:258-322
    return _ctx(problem).approximate_model(u_array, x_array)


def initialize_(ileqg: ILEQGSolver, problem, x_0, u_array, theta):          # initialize!  ileqg.jl:214-236
    ctx = ileqg.context(problem)
    ileqg.mu, ileqg.Delta = 0.0, ileqg.Delta_0
    ileqg.d_current, ileqg.iter_current = np.inf, 0
    ileqg.eps_init = ileqg.eps_init_init
    ileqg.eps_history = []
    ileqg.x_array = ctx.rollout_open(x_0, u_array)
    ileqg.l_array = np.array(u_array, dtype=np.float64)
    ileqg.L_array = np.zeros((problem.N, problem.m, problem.n))
    ap = ctx.approximate_model(ileqg.l_array, ileqg.x_array)
    st, dp = ctx.dp_policy_eval(ap, ileqg.L_array, None, theta, ileqg.mu)
    assert st == 0, "M: (inv(W) - θ*S) is not PSD"                            # the @assert at :440
    ileqg.value_current = dp.s_array[0]


def solve_approximate_dp_(ileqg: ILEQGSolver, approx_result, verbose=False, theta=0.0):   # solve_approximate_dp!  :341-406
    st, L, dl, dp, mu, delta = ileqg.ctx.dp_gain_sweep(approx_result, theta, ileqg.mu, ileqg.Delta)
    ileqg.mu, ileqg.Delta = mu, delta
    assert st != nv.ST_M_NOT_PD_GAIN, "M: (inv(W) - θ*S) is not PSD"           # the @assert at :366
    if st != 0:
        raise ArithmeticError(f"solve_approximate_dp!: status {st}")
    ileqg.L_array = L
    return dp, dl


def solve_approximate_dp(approx_result, L_array, dl_array=None, theta=0.0, mu=0.0, ctx: Context | None = None,
                         problem=None):                                         # ileqg.jl:412-465
    if ctx is None:
        if problem is None:
            raise ValueError("solve_approximate_dp needs ctx= or problem= (the W(k) tables live in the problem)")
        ctx = _ctx(problem)
    st, dp = ctx.dp_policy_eval(approx_result, L_array, dl_array, theta, mu)
    assert st == 0, "M: (inv(W) - θ*S) is not PSD"
    return dp


def increase_mu_and_delta_(ileqg: ILEQGSolver):            # increase_μ_and_Δ!  ileqg.jl:471-474
    ileqg.Delta = max(ileqg.Delta_0, ileqg.Delta * ileqg.Delta_0)
    ileqg.mu = max(ileqg.mu_min, ileqg.mu * ileqg.Delta)


def decrease_mu_and_delta_(ileqg: ILEQGSolver):            # decrease_μ_and_Δ!  ileqg.jl:480-488
    ileqg.Delta = min(1 / ileqg.Delta_0, ileqg.Delta / ileqg.Delta_0)
    cand = ileqg.mu * ileqg.Delta
    ileqg.mu = cand if cand >= ileqg.mu_min else 0.0


def _isapprox(x, y):
    if x == y:
        return True
    if not (np.isfinite(x) and np.isfinite(y)):
        return False
    return abs(x - y) <= SQRT_EPS * max(abs(x), abs(y))


def line_search_(ileqg: ILEQGSolver, problem, dl_array_new, theta, verbose=False):      # line_search!  ileqg.jl:494-592
    ctx = ileqg.context(problem)
    cur = ileqg.value_current
    eps = ileqg.eps_init
    count = 0
    dl_array_new = np.asarray(dl_array_new, dtype=np.float64)
    while True:
        count += 1
        l_new = ileqg.l_array + eps * dl_array_new                                            # :509
        x_new, u_new = ctx.rollout_feedback(ileqg.x_array, l_new, ileqg.L_array)             # :517
        ap_new = ctx.approximate_model(u_new, x_new)                                           # :520
        st, dp_new = ctx.dp_policy_eval(ap_new, ileqg.L_array, None, theta, ileqg.mu)          # :522-528
        if st != 0:
            eps *= ileqg.lam                                                                   # :529-535
            continue
        new = dp_new.s_array[0]
        ileqg.eps_history.append((eps, new - cur))                                             # :537
        accept = _isapprox(new, cur) or new < cur                                              # :538
        if not accept:
            eps *= ileqg.lam                                                                   # :557
            if not eps < ileqg.eps_min:
                continue
        ileqg.d_current = float(np.max(np.linalg.norm(ileqg.l_array - u_new, axis=1)))        # :539 / :559
        ileqg.value_current = new
        ileqg.x_array, ileqg.l_array = x_new, u_new
        break
    if ileqg.eps_init_auto:                                                                    # :582-591
        if count == 1:
            ileqg.eps_init = min(ileqg.eps_init_init, eps / ileqg.lam)
        else:
            while eps < ileqg.eps_min:
                eps = eps / ileqg.lam
            ileqg.eps_init = eps


def step_(ileqg: ILEQGSolver, problem, theta, verbose=False):                 # step!  ileqg.jl:598-613
    ileqg.iter_current += 1
    ap = ileqg.context(problem).approximate_model(ileqg.l_array, ileqg.x_array)            # :604
    _, dl = solve_approximate_dp_(ileqg, ap, verbose, theta=theta)            # :610-611
    line_search_(ileqg, problem, dl, theta, verbose)                          # :612


def solve_(ileqg: ILEQGSolver, problem, x_0, u_array, theta, verbose=False):
    """solve!(ileqg, problem, x_0, u_array; θ)  (ileqg.jl:635-659) on the fused device state machine.

    Returns (x_array, l_array, L_array, value, ϵ_history).  Raises where the reference throws."""
    if getattr(problem, "model", 0) == 0:            # generic closures: host rollouts + linearisation, device sweeps
        return solve_stepwise_(ileqg, problem, x_0, u_array, theta)
    r = ileqg.context(problem).solve(x_0, u_array, theta)
    ileqg.status = r["status"]
    ileqg.iter_current = r["iters"]
    ileqg.eps_history = [tuple(p) for p in r["eps_history"]]
    if r["status"] in (nv.ST_M_NOT_PD_INIT, nv.ST_M_NOT_PD_GAIN):
        raise AssertionError("M: (inv(W) - θ*S) is not PSD")
    if r["status"] not in (nv.ST_OK, nv.ST_ITER_MAX):
        raise ArithmeticError(f"iLEQG solve failed with status {r['status']}")
    ileqg.x_array, ileqg.l_array, ileqg.L_array, ileqg.value_current = r["x"], r["l"], r["L"], r["value"]
    return r["x"].copy(), r["l"].copy(), r["L"].copy(), r["value"], list(ileqg.eps_history)


def solve_stepwise_(ileqg: ILEQGSolver, problem, x_0, u_array, theta):
    """The same solve!, composed from initialize_/step_ through the operator entry points (test aid)."""
    initialize_(ileqg, problem, x_0, u_array, theta)
    while True:
        step_(ileqg, problem, theta)
        if ileqg.d > ileqg.d_current and ileqg.mu <= ileqg.mu_min:
            break
        elif ileqg.iter_current == ileqg.iter_max:
            break
    return ileqg.x_array.copy(), ileqg.l_array.copy(), ileqg.L_array.copy(), ileqg.value_current, list(ileqg.eps_history)
