"""ctypes front-end of the CPU ORACLE (oracle/libratilqr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (ratilqr.jl_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libratilqr_oracle.so")

OK, ERR_M_NOT_PD_INIT, ERR_M_NOT_PD_GAIN, ITER_MAX, ERR_DOMAIN, ERR_MU_DIVERGED, ERR_SINGULAR = range(7)

_dp = C.POINTER(C.c_double)


class _Problem(C.Structure):
    _fields_ = [("model", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32),
                ("cost_tv", C.c_int32), ("W_tv", C.c_int32),
                ("A", _dp), ("B", _dp), ("Q", _dp), ("R", _dp), ("P", _dp), ("qv", _dp), ("rv", _dp), ("q0", _dp),
                ("Qf", _dp), ("qvf", _dp), ("q0f", C.c_double), ("kappa", C.c_double),
                ("pl_a", C.c_double), ("pl_b", C.c_double), ("pl_p", C.c_double), ("pl_pu", C.c_double),
                ("pl_cx", C.c_double), ("pl_cu", C.c_double), ("pl_h", C.c_double),
                ("W", _dp)]


class _Opts(C.Structure):
    _fields_ = [("mu_min", C.c_double), ("delta_0", C.c_double), ("lam", C.c_double), ("d", C.c_double),
                ("iter_max", C.c_int64), ("eps_init", C.c_double), ("eps_min", C.c_double),
                ("adaptive_eps_init", C.c_int32)]


class _Approx(C.Structure):
    _fields_ = [(k, _dp) for k in ("q", "qv", "Q", "r", "R", "P", "A", "B", "W")]


class _Dp(C.Structure):
    _fields_ = [(k, _dp) for k in ("s", "sv", "S", "g", "G", "H")]


class _Solver(C.Structure):
    _fields_ = [("o", _Opts), ("mu", C.c_double), ("delta", C.c_double), ("eps_init_cur", C.c_double),
                ("value_current", C.c_double), ("d_current", C.c_double), ("iter_current", C.c_int64),
                ("n", C.c_int), ("m", C.c_int), ("N", C.c_int),
                ("x", _dp), ("l", _dp), ("L", _dp), ("eps_hist", _dp), ("n_hist", C.c_int64), ("cap_hist", C.c_int64),
                ("n_ls_evals", C.c_int64), ("ap", C.POINTER(_Approx)), ("ap_new", C.POINTER(_Approx)),
                ("dp", C.POINTER(_Dp)), ("dl", _dp), ("x_new", _dp), ("u_new", _dp), ("l_new", _dp)]


class _Ce(C.Structure):
    _fields_ = [("ileqg", _Opts), ("num_samples", C.c_int64), ("num_elite", C.c_int64), ("iter_max", C.c_int64),
                ("lam", C.c_double), ("use_theta_max", C.c_int32),
                ("mu_init", C.c_double), ("sigma_init", C.c_double), ("mu", C.c_double), ("sigma", C.c_double),
                ("theta_max", C.c_double), ("theta_min", C.c_double), ("iter_current", C.c_int64),
                ("z", _dp), ("nz", C.c_int64), ("zpos", C.c_int64),
                ("n_solves", C.c_int64), ("n_redraws", C.c_int64), ("nthreads", C.c_int),
                ("n_final_retries", C.c_int64)]


_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ratilqr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.environ.get("RATILQR_ORACLE_SO", _SO))          # override: the ASan / UBSan build of tools/sanitize_host.sh
        _lib.orc_solver_new.restype = C.POINTER(_Solver)
        _lib.orc_approx_alloc.restype = C.POINTER(_Approx)
        _lib.orc_dp_alloc.restype = C.POINTER(_Dp)
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _cm2(a):  # (rows, cols) -> flat column-major
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).ravel()


def _cm3(a):  # (T, rows, cols) -> flat, time slowest, column-major inside
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).transpose(0, 2, 1)).ravel()


def _from_cm3(buf, T, rows, cols):
    return np.array(buf, dtype=np.float64).reshape(T, cols, rows).transpose(0, 2, 1).copy()


class Problem:
    """Owns the ctypes view of a ratilqr.jl_amd problem object (anything with .c_tables())."""

    def __init__(self, prob):
        t = prob.c_tables()
        self.n, self.m, self.N = t["n"], t["m"], t["N"]
        self._keep = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in t.items() if isinstance(v, np.ndarray)}
        s = _Problem()
        for k in ("model", "n", "m", "N", "cost_tv", "W_tv"):
            setattr(s, k, int(t[k]))
        for k in ("q0f", "kappa", "pl_a", "pl_b", "pl_p", "pl_pu", "pl_cx", "pl_cu", "pl_h"):
            setattr(s, k, float(t[k]))
        for k, v in self._keep.items():
            setattr(s, k, _p(v))
        self.c = s


def make_opts(**kw) -> _Opts:
    o = _Opts()
    lib().orc_default_opts(C.byref(o))
    ren = {"lambda_": "lam", "lam": "lam", "adaptive_eps_init": "adaptive_eps_init"}
    for k, v in kw.items():
        setattr(o, ren.get(k, k), v)
    return o


# ---- stateless ops ---------------------------------------------------------------------------
def simulate_open(P: Problem, x0, u):
    x = np.zeros((P.N + 1, P.n))
    u = np.ascontiguousarray(u, dtype=np.float64)
    rc = lib().orc_simulate_open(C.byref(P.c), _p(np.ascontiguousarray(x0, dtype=np.float64)), _p(u), _p(x))
    return rc, x


def simulate_feedback(P: Problem, xbar, l, L):
    xn, un = np.zeros((P.N + 1, P.n)), np.zeros((P.N, P.m))
    xbar, l, Lc = np.ascontiguousarray(xbar, float), np.ascontiguousarray(l, float), _cm3(L)
    rc = lib().orc_simulate_feedback(C.byref(P.c), _p(xbar), _p(l), _p(Lc), _p(xn), _p(un))
    return rc, xn, un


def simulate_noisy(P: Problem, x_nom, l, L, z):
    """K noisy rollouts (open loop if L is None); z has shape (K, N, n).  Returns rc, x (K, N+1, n), u (K, N, m), cost (K,)."""
    z = np.ascontiguousarray(z, float)
    K = z.shape[0]
    x, u, cost = np.zeros((K, P.N + 1, P.n)), np.zeros((K, P.N, P.m)), np.zeros(K)
    x_nom, l = np.ascontiguousarray(x_nom, float), np.ascontiguousarray(l, float)
    Lc = _cm3(L) if L is not None else None
    rc = lib().orc_simulate_noisy(C.byref(P.c), _p(x_nom), _p(l), _p(Lc) if Lc is not None else None, C.c_int64(K),
                                  _p(z), _p(x), _p(u), _p(cost))
    return rc, x, u, cost


def integrate_cost(P: Problem, x, u):
    out = C.c_double()
    x, u = np.ascontiguousarray(x, float), np.ascontiguousarray(u, float)
    rc = lib().orc_integrate_cost(C.byref(P.c), _p(x), _p(u), C.byref(out))
    return rc, out.value


class Approx:
    """ApproximationResult (ileqg.jl:242-252) as numpy arrays with natural (time, row, col) shapes."""

    def __init__(self, P: Problem):
        self.P = P
        self.ptr = lib().orc_approx_alloc(P.n, P.m, P.N)

    def __del__(self):
        try:
            lib().orc_approx_free(self.ptr)
        except Exception:
            pass

    def arrays(self):
        n, m, N = self.P.n, self.P.m, self.P.N
        a = self.ptr.contents
        g = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
        return dict(q=g(a.q, N + 1), qv=g(a.qv, n * (N + 1)).reshape(N + 1, n),
                    Q=_from_cm3(g(a.Q, n * n * (N + 1)), N + 1, n, n), r=g(a.r, m * N).reshape(N, m),
                    R=_from_cm3(g(a.R, m * m * N), N, m, m), P=_from_cm3(g(a.P, m * n * N), N, m, n),
                    A=_from_cm3(g(a.A, n * n * N), N, n, n), B=_from_cm3(g(a.B, n * m * N), N, n, m),
                    W=_from_cm3(g(a.W, n * n * N), N, n, n))


def approximate_model(P: Problem, u, x):
    ap = Approx(P)
    u, x = np.ascontiguousarray(u, float), np.ascontiguousarray(x, float)
    rc = lib().orc_approximate_model(C.byref(P.c), _p(u), _p(x), ap.ptr)
    return rc, ap


def _dp_arrays(P, d):
    n, m, N = P.n, P.m, P.N
    g = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy()
    return dict(s=g(d.s, N + 1), sv=g(d.sv, n * (N + 1)).reshape(N + 1, n),
                S=_from_cm3(g(d.S, n * n * (N + 1)), N + 1, n, n), g=g(d.g, m * N).reshape(N, m),
                G=_from_cm3(g(d.G, m * n * N), N, m, n), H=_from_cm3(g(d.H, m * m * N), N, m, m))


def dp_gain(P: Problem, ap: Approx, theta, mu=0.0, delta=2.0, mu_min=1e-6, delta_0=2.0):
    """solve_approximate_dp! -> (rc, L (N,m,n), dl (N,m), dp dict, mu, delta)."""
    n, m, N = P.n, P.m, P.N
    Lb, dl = np.zeros(m * n * N), np.zeros((N, m))
    mu_c, de_c = C.c_double(mu), C.c_double(delta)
    d = lib().orc_dp_alloc(n, m, N)
    rc = lib().orc_dp_gain(n, m, N, ap.ptr, C.c_double(theta), C.c_double(mu_min), C.c_double(delta_0),
                           C.byref(mu_c), C.byref(de_c), _p(Lb), _p(dl), d)
    out = _dp_arrays(P, d.contents)
    lib().orc_dp_free(d)
    return rc, _from_cm3(Lb, N, m, n), dl, out, mu_c.value, de_c.value


def dp_eval(P: Problem, ap: Approx, L, dl, theta, mu):
    n, m, N = P.n, P.m, P.N
    Lc = _cm3(L)
    dlc = None if dl is None else np.ascontiguousarray(dl, float)
    d = lib().orc_dp_alloc(n, m, N)
    rc = lib().orc_dp_eval(n, m, N, ap.ptr, _p(Lc), _p(dlc) if dlc is not None else None,
                           C.c_double(theta), C.c_double(mu), d)
    out = _dp_arrays(P, d.contents)
    lib().orc_dp_free(d)
    return rc, out


# ---- ILEQGSolver --------------------------------------------------------------------------------
class ILEQGSolver:
    def __init__(self, P: Problem, **kw):
        self.P = P
        self.opts = make_opts(**kw)
        self.ptr = lib().orc_solver_new(C.byref(P.c), C.byref(self.opts))
        if not self.ptr:
            raise AssertionError("invalid ILEQGSolver options (ileqg.jl:195-201)")

    def __del__(self):
        try:
            lib().orc_solver_free(self.ptr)
        except Exception:
            pass

    @property
    def s(self):
        return self.ptr.contents

    def _arr(self, p, k):
        return np.ctypeslib.as_array(p, shape=(k,)).copy()

    @property
    def x_array(self):
        return self._arr(self.s.x, self.P.n * (self.P.N + 1)).reshape(self.P.N + 1, self.P.n)

    @property
    def l_array(self):
        return self._arr(self.s.l, self.P.m * self.P.N).reshape(self.P.N, self.P.m)

    @property
    def L_array(self):
        return _from_cm3(self._arr(self.s.L, self.P.m * self.P.n * self.P.N), self.P.N, self.P.m, self.P.n)

    @property
    def dl_array(self):
        return self._arr(self.s.dl, self.P.m * self.P.N).reshape(self.P.N, self.P.m)

    @property
    def eps_history(self):
        k = self.s.n_hist
        return self._arr(self.s.eps_hist, 2 * k).reshape(k, 2) if k else np.zeros((0, 2))

    def initialize(self, x0, u, theta):
        return lib().orc_initialize(self.ptr, C.byref(self.P.c), _p(np.ascontiguousarray(x0, float)),
                                    _p(np.ascontiguousarray(u, float)), C.c_double(theta))

    def set_L(self, L):
        Lc = _cm3(L)
        C.memmove(self.s.L, Lc.ctypes.data, Lc.nbytes)

    def line_search(self, dl, theta):
        return lib().orc_line_search(self.ptr, C.byref(self.P.c), _p(np.ascontiguousarray(dl, float)), C.c_double(theta))

    def step(self, theta):
        return lib().orc_step(self.ptr, C.byref(self.P.c), C.c_double(theta))

    def solve(self, x0, u, theta):
        return lib().orc_solve(self.ptr, C.byref(self.P.c), _p(np.ascontiguousarray(x0, float)),
                               _p(np.ascontiguousarray(u, float)), C.c_double(theta))

    def increase_mu_delta(self):
        lib().orc_increase_mu_delta(self.ptr)

    def decrease_mu_delta(self):
        lib().orc_decrease_mu_delta(self.ptr)


def compute_value_batch(P: Problem, x0, u, theta, nthreads=1, **opts):
    theta = np.ascontiguousarray(theta, float)
    B = theta.size
    value, status = np.zeros(B), np.zeros(B, np.int32)
    iters, ls = np.zeros(B, np.int32), np.zeros(B, np.int32)
    o = make_opts(**opts)
    lib().orc_compute_value_batch(C.byref(P.c), C.byref(o), _p(np.ascontiguousarray(x0, float)),
                                  _p(np.ascontiguousarray(u, float)), _p(theta), C.c_int64(B), _p(value),
                                  status.ctypes.data_as(C.POINTER(C.c_int32)), iters.ctypes.data_as(C.POINTER(C.c_int32)),
                                  ls.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(nthreads))
    return value, status, iters, ls


# ---- CrossEntropyBilevelOptimizationSolver --------------------------------------------------------
class CrossEntropyBilevelOptimizationSolver:
    def __init__(self, z_stream, num_samples=10, num_elite=3, iter_max=5, lam=0.5, mu_init=1.0, sigma_init=2.0,
                 use_theta_max=False, nthreads=1, **ileqg_opts):
        self.c = _Ce()
        lib().orc_ce_default(C.byref(self.c))
        self.c.ileqg = make_opts(**ileqg_opts)
        self.c.num_samples, self.c.num_elite, self.c.iter_max = num_samples, num_elite, iter_max
        self.c.lam, self.c.mu_init, self.c.sigma_init = lam, mu_init, sigma_init
        self.c.mu, self.c.sigma = mu_init, sigma_init
        self.c.use_theta_max = int(use_theta_max)
        self.c.nthreads = nthreads
        self._z = np.ascontiguousarray(z_stream, float)
        self.c.z, self.c.nz, self.c.zpos = _p(self._z), self._z.size, 0

    def initialize(self):
        lib().orc_ce_initialize(C.byref(self.c))

    def get_positive_samples(self, mu, sigma, num):
        th = np.zeros(num)
        rc = lib().orc_ce_get_positive_samples(C.byref(self.c), C.c_double(mu), C.c_double(sigma), C.c_int64(num), _p(th))
        return rc, th

    def elite_update(self, theta, cost):
        """the tail of step! (:314-334) on given thetas / costs"""
        th, co = np.ascontiguousarray(theta, float), np.ascontiguousarray(cost, float)
        assert th.size == co.size == self.c.num_samples
        lib().orc_ce_elite_update(C.byref(self.c), _p(th), _p(co))

    def step(self, P: Problem, x0, u, kl_bound):
        B = self.c.num_samples
        th, cost = np.zeros(B), np.zeros(B)
        rc = lib().orc_ce_step(C.byref(self.c), C.byref(P.c), _p(np.ascontiguousarray(x0, float)),
                               _p(np.ascontiguousarray(u, float)), C.c_double(kl_bound), _p(th), _p(cost))
        return rc, th, cost

    def solve(self, P: Problem, x0, u, kl_bound):
        n, m, N = P.n, P.m, P.N
        x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
        th, val, tmin, tmax = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        rc = lib().orc_ce_solve(C.byref(self.c), C.byref(P.c), _p(np.ascontiguousarray(x0, float)),
                                _p(np.ascontiguousarray(u, float)), C.c_double(kl_bound), C.byref(th), _p(x), _p(l),
                                _p(Lb), C.byref(val), C.byref(tmin), C.byref(tmax))
        return rc, th.value, x, l, _from_cm3(Lb, N, m, n), val.value, tmin.value, tmax.value


# ---- NelderMeadBilevelOptimizationSolver (RAT iLQR++) ---------------------------------------------
class _Nm(C.Structure):
    _fields_ = [("ileqg", _Opts), ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double), ("eps", C.c_double),
                ("lam", C.c_double), ("iter_max", C.c_int64), ("theta_high_init", C.c_double), ("theta_low_init", C.c_double),
                ("iter_current", C.c_int64), ("theta_high", C.c_double), ("theta_low", C.c_double),
                ("has_c_high", C.c_int32), ("has_c_low", C.c_int32), ("c_high", C.c_double), ("c_low", C.c_double),
                ("n_solves", C.c_int64)]


class NelderMeadBilevelOptimizationSolver:
    def __init__(self, alpha=1.0, beta=2.0, gamma=0.5, eps=1e-2, lam=0.5, iter_max=100, theta_high_init=3.0,
                 theta_low_init=1e-8, **ileqg_opts):
        self.c = _Nm()
        lib().orc_nm_default(C.byref(self.c))
        self.c.ileqg = make_opts(**ileqg_opts)
        self.c.alpha, self.c.beta, self.c.gamma, self.c.eps, self.c.lam = alpha, beta, gamma, eps, lam
        self.c.iter_max = iter_max
        self.c.theta_high_init = self.c.theta_high = theta_high_init
        self.c.theta_low_init = self.c.theta_low = theta_low_init

    def initialize(self):
        lib().orc_nm_initialize(C.byref(self.c))

    def compute_cost(self, P: Problem, x0, u, theta, kl_bound):
        f = lib().orc_nm_compute_cost
        f.restype = C.c_double
        return f(C.byref(self.c), C.byref(P.c), _p(np.ascontiguousarray(x0, float)), _p(np.ascontiguousarray(u, float)),
                 C.c_double(theta), C.c_double(kl_bound))

    def step(self, P: Problem, x0, u, kl_bound):
        lib().orc_nm_step(C.byref(self.c), C.byref(P.c), _p(np.ascontiguousarray(x0, float)),
                          _p(np.ascontiguousarray(u, float)), C.c_double(kl_bound))

    def solve(self, P: Problem, x0, u, kl_bound):
        n, m, N = P.n, P.m, P.N
        x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
        th, val = C.c_double(), C.c_double()
        rc = lib().orc_nm_solve(C.byref(self.c), C.byref(P.c), _p(np.ascontiguousarray(x0, float)),
                                _p(np.ascontiguousarray(u, float)), C.c_double(kl_bound), C.byref(th), _p(x), _p(l), _p(Lb),
                                C.byref(val))
        return rc, th.value, x, l, _from_cm3(Lb, N, m, n), val.value


# ---- PETS (pets.jl) on the generative family -------------------------------------------------------------
class _GenProblem(C.Structure):
    _fields_ = [("lq", _Problem), ("l1u", C.c_double), ("noise_kind", C.c_int32), ("nmean", _dp), ("nchol", _dp),
                ("nlo", C.c_double), ("nhi", C.c_double), ("tw2", C.c_double), ("tmean2", _dp), ("tchol2", _dp)]


class _Pets(C.Structure):
    _fields_ = [("num_control_samples", C.c_int64), ("num_trajectory_samples", C.c_int64), ("num_elite", C.c_int64),
                ("iter_max", C.c_int64), ("smoothing_factor", C.c_double), ("N", C.c_int64), ("m", C.c_int64),
                ("iter_current", C.c_int64), ("mu_init", _dp), ("Sigma_init", _dp), ("mu", _dp), ("Sigma", _dp)]


class GenProblem:
    def __init__(self, prob):
        self.lq = Problem(prob.lq)
        self.n, self.m, self.N = prob.n, prob.m, prob.N
        t = prob.gen_tables()
        self._keep = {k: np.ascontiguousarray(t[k], dtype=np.float64) for k in ("nmean", "nchol", "tmean2", "tchol2")}
        g = _GenProblem()
        g.lq = self.lq.c
        g.l1u, g.noise_kind, g.nlo, g.nhi, g.tw2 = t["l1u"], t["noise_kind"], t["nlo"], t["nhi"], t["tw2"]
        for k, v in self._keep.items():
            setattr(g, k, _p(v))
        self.c = g


def pets_compute_cost(G: GenProblem, x0, controls, K, use_true_model, zn, zu=None):
    controls = np.ascontiguousarray(controls, float)
    S = controls.shape[0]
    cost = np.zeros(S)
    zn = np.ascontiguousarray(zn, float)
    zu = None if zu is None else np.ascontiguousarray(zu, float)
    lib().orc_pets_compute_cost(C.byref(G.c), _p(np.ascontiguousarray(x0, float)), _p(controls), C.c_int64(S), C.c_int64(K),
                                C.c_int(int(use_true_model)), _p(zn), _p(zu) if zu is not None else None, _p(cost))
    return cost


class PetsSolver:
    def __init__(self, mu_init, Sigma_init, num_control_samples=10, num_trajectory_samples=10, num_elite=3, iter_max=5,
                 smoothing_factor=0.1):
        mu_init, Sigma_init = np.asarray(mu_init, float), np.asarray(Sigma_init, float)
        self.N, self.m = mu_init.shape
        self._mi, self._si = np.ascontiguousarray(mu_init).ravel().copy(), _cm3(Sigma_init).copy()
        self._mu, self._sg = self._mi.copy(), self._si.copy()
        c = _Pets()
        c.num_control_samples, c.num_trajectory_samples, c.num_elite, c.iter_max = num_control_samples, num_trajectory_samples, num_elite, iter_max
        c.smoothing_factor, c.N, c.m, c.iter_current = smoothing_factor, self.N, self.m, 0
        c.mu_init, c.Sigma_init, c.mu, c.Sigma = _p(self._mi), _p(self._si), _p(self._mu), _p(self._sg)
        self.c = c

    mu_array = property(lambda s: s._mu.reshape(s.N, s.m).copy())
    Sigma_array = property(lambda s: s._sg.reshape(s.N, s.m, s.m).transpose(0, 2, 1).copy())

    def initialize(self):
        lib().orc_pets_initialize(C.byref(self.c))

    def update(self, controls, cost):
        idx = np.zeros(self.c.num_elite, np.int64)
        lib().orc_pets_update(C.byref(self.c), _p(np.ascontiguousarray(controls, float)), _p(np.ascontiguousarray(cost, float)),
                              idx.ctypes.data_as(C.POINTER(C.c_int64)))
        return idx

    def step(self, G: GenProblem, x0, use_true_model, zc, zn, zu=None):
        S = self.c.num_control_samples
        ctrl, cost = np.zeros((S, self.N, self.m)), np.zeros(S)
        zc, zn = np.ascontiguousarray(zc, float), np.ascontiguousarray(zn, float)
        zu = None if zu is None else np.ascontiguousarray(zu, float)
        rc = lib().orc_pets_step(C.byref(self.c), C.byref(G.c), _p(np.ascontiguousarray(x0, float)), C.c_int(int(use_true_model)),
                                 _p(zc), _p(zn), _p(zu) if zu is not None else None, _p(ctrl), _p(cost))
        return rc, ctrl, cost


# ---- generic closures: the reference's own solve! loop over host closures, sweeps by the C oracle ----------------
class ClosureProblem:
    """FiniteHorizonRiskSensitiveOptimalControlProblem(f, c, h, W, N) with arbitrary Python closures (optimal_control_problems.jl:67-73).
    jac(x, u) -> (A, B), cost_derivs(k, x, u) -> (q_vec, Q, r_vec, R, P), term_derivs(x) -> (q_vec, Q) stand for the ForwardDiff closures
    of ileqg.jl:265-273 (exact derivatives supplied by the test)."""

    def __init__(self, f, c, h, W, N, n, m, jac, cost_derivs, term_derivs):
        self.f, self.c, self.h, self.W, self.N, self.n, self.m = f, c, h, W, int(N), int(n), int(m)
        self.jac, self.cost_derivs, self.term_derivs = jac, cost_derivs, term_derivs


def approx_from_arrays(shape, q, qv, Q, r, R, P, A, B, W) -> Approx:
    """An orc_approx filled from ApproximationResult arrays of natural (time, row, col) shapes; `shape` has n, m, N."""
    ap = Approx(shape)
    a = ap.ptr.contents
    for dst, src in ((a.q, np.ascontiguousarray(q, float)), (a.qv, np.ascontiguousarray(qv, float).ravel()), (a.Q, _cm3(Q)),
                     (a.r, np.ascontiguousarray(r, float).ravel()), (a.R, _cm3(R)), (a.P, _cm3(P)), (a.A, _cm3(A)), (a.B, _cm3(B)),
                     (a.W, _cm3(W))):
        C.memmove(dst, src.ctypes.data, src.nbytes)
    return ap


def closure_approximate_model(cp: ClosureProblem, u, x) -> Approx:                      # approximate_model  ileqg.jl:258-322
    n, m, N = cp.n, cp.m, cp.N
    q, qv, Q = np.zeros(N + 1), np.zeros((N + 1, n)), np.zeros((N + 1, n, n))
    r, R, P = np.zeros((N, m)), np.zeros((N, m, m)), np.zeros((N, m, n))
    A, B, W = np.zeros((N, n, n)), np.zeros((N, n, m)), np.zeros((N, n, n))
    for k in range(N):                                                                   # k is the reference's 0-based time (:284-313)
        q[k] = cp.c(k, x[k], u[k])
        qv[k], Q[k], r[k], R[k], P[k] = cp.cost_derivs(k, x[k], u[k])
        A[k], B[k] = cp.jac(x[k], u[k])
        W[k] = cp.W(k)
    q[N] = cp.h(x[N])
    qv[N], Q[N] = cp.term_derivs(x[N])
    return approx_from_arrays(cp, q, qv, Q, r, R, P, A, B, W)


def closure_solve(cp: ClosureProblem, x0, u0, theta, mu_min=1e-6, delta_0=2.0, lam=0.5, d=1e-2, iter_max=100, eps_init=1.0, eps_min=1e-6,
                  adaptive_eps_init=False):
    """solve!(ileqg, problem, x_0, u_array; theta) (ileqg.jl:635-659) statement by statement over host closures:
    initialize! (:214-236), step! (:598-613), line_search! (:494-592); the sweeps are orc_dp_gain / orc_dp_eval.
    Returns dict(status, value, x, l, L, iters, ls_evals, eps_history)."""
    n, m, N = cp.n, cp.m, cp.N
    sqrt_eps = 1.4901161193847656e-8

    def rollout_open(x0, u):
        x = np.zeros((N + 1, n)); x[0] = x0
        for t in range(N):
            x[t + 1] = cp.f(x[t], u[t])
        return x

    def rollout_fb(xbar, l, L):
        xn, un = np.zeros((N + 1, n)), np.zeros((N, m)); xn[0] = xbar[0]
        for t in range(N):
            un[t] = l[t] + L[t] @ (xn[t] - xbar[t])
            xn[t + 1] = cp.f(xn[t], un[t])
        return xn, un

    def isapprox(a, b):
        if a == b:
            return True
        if not (np.isfinite(a) and np.isfinite(b)):
            return False
        return abs(a - b) <= sqrt_eps * max(abs(a), abs(b))

    out = dict(status=0, value=np.inf, iters=0, ls_evals=0, eps_history=[])
    mu, delta, d_cur, eps_i = 0.0, delta_0, np.inf, eps_init                              # :216-219
    x = rollout_open(np.asarray(x0, float), np.asarray(u0, float))                        # :225
    l, L = np.array(u0, float), np.zeros((N, m, n))                                       # :228-232
    rc, dp = dp_eval(cp, closure_approximate_model(cp, l, x), L, None, theta, mu)         # :233-234
    if rc:
        out["status"] = 1
        return out
    value = dp["s"][0]
    it = 0
    while True:
        it += 1                                                                           # step!  :599
        rc, L, dl, _, mu, delta = dp_gain(cp, closure_approximate_model(cp, l, x), theta, mu, delta, mu_min, delta_0)   # :604-611
        if rc:
            out.update(status=2 if rc == 2 else 5, iters=it)
            return out
        eps, count = eps_i, 0                                                             # line_search!  :502
        while True:
            count += 1
            out["ls_evals"] += 1
            if count > 4000:
                out.update(status=7, iters=it)
                return out
            xn, un = rollout_fb(x, l + eps * dl, L)                                       # :509-519
            rc, dpn = dp_eval(cp, closure_approximate_model(cp, un, xn), L, None, theta, mu)   # :520-528
            if rc:
                eps *= lam                                                                # :529-535
                continue
            new = dpn["s"][0]
            out["eps_history"].append((eps, new - value))                                 # :537
            if not (isapprox(new, value) or new < value):                                 # :538
                eps *= lam                                                                # :557
                if not eps < eps_min:
                    continue
            d_cur = float(np.max(np.linalg.norm(l - un, axis=1)))                         # :539 / :559
            value, x, l = new, xn, un
            break
        if adaptive_eps_init:                                                             # :582-591
            if count == 1:
                eps_i = min(eps_init, eps / lam)
            else:
                while eps < eps_min:
                    eps = eps / lam
                eps_i = eps
        if d > d_cur and mu <= mu_min:                                                    # :642
            break
        if it == iter_max:                                                                # :648
            out["status"] = 3
            break
    out.update(value=value, x=x, l=l, L=L, iters=it)
    return out
