"""ctypes binding of csrc/libratilqr_hip.so (C ABI: include/ratilqr.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is visible, every
compute entry point raises.  (The CPU oracle lives under oracle/ and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RATILQR_SO", os.path.join(_HERE, "csrc", "libratilqr_hip.so"))   # override: diagnostic builds only

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

RC_NAMES = {0: "RAT_OK", 1: "RAT_ERR_ARG", 2: "RAT_ERR_UNSUPPORTED", 3: "RAT_ERR_HIP", 4: "RAT_ERR_NO_PROBLEM",
            5: "RAT_ERR_STREAM_DRY", 6: "RAT_ERR_DIVERGED"}
ST_RUNNING, ST_OK, ST_M_NOT_PD_INIT, ST_M_NOT_PD_GAIN, ST_ITER_MAX, ST_DOMAIN, ST_MU_DIVERGED, ST_SINGULAR, \
    ST_LS_DIVERGED = -1, 0, 1, 2, 3, 4, 5, 6, 7
K_NAMES = ("rollout", "linearize", "sweep_eval", "sweep_gain", "select", "sweep_init", "sweep_dual", "solve_fused", "solve_block", "solve_wide",
           "pets", "ce_bookkeeping")


class RatError(RuntimeError):
    pass


class ProblemDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32),
                ("cost_tv", C.c_int32), ("W_tv", C.c_int32),
                ("A", _dp), ("B", _dp), ("Q", _dp), ("R", _dp), ("P", _dp), ("qv", _dp), ("rv", _dp), ("q0", _dp),
                ("Qf", _dp), ("qvf", _dp), ("q0f", C.c_double), ("kappa", C.c_double),
                ("pl_a", C.c_double), ("pl_b", C.c_double), ("pl_p", C.c_double), ("pl_pu", C.c_double),
                ("pl_cx", C.c_double), ("pl_cu", C.c_double), ("pl_h", C.c_double),
                ("W", _dp)]


class IleqgOpts(C.Structure):
    _fields_ = [("mu_min", C.c_double), ("delta_0", C.c_double), ("lam", C.c_double), ("d", C.c_double),
                ("iter_max", C.c_int64), ("eps_init", C.c_double), ("eps_min", C.c_double),
                ("adaptive_eps_init", C.c_int32)]


class CeSolver(C.Structure):
    _fields_ = [("num_samples", C.c_int64), ("num_elite", C.c_int64), ("iter_max", C.c_int64),
                ("lam", C.c_double), ("use_theta_max", C.c_int32),
                ("mu_init", C.c_double), ("sigma_init", C.c_double), ("mu", C.c_double), ("sigma", C.c_double),
                ("theta_max", C.c_double), ("theta_min", C.c_double), ("iter_current", C.c_int64),
                ("n_solves", C.c_int64), ("n_redraws", C.c_int64), ("n_final_retries", C.c_int64)]


class GenProblemDesc(C.Structure):
    _fields_ = [("lq", ProblemDesc), ("l1u", C.c_double), ("noise_kind", C.c_int32), ("nmean", _dp), ("nchol", _dp),
                ("nlo", C.c_double), ("nhi", C.c_double), ("tw2", C.c_double), ("tmean2", _dp), ("tchol2", _dp)]


class PetsSolver(C.Structure):
    _fields_ = [("num_control_samples", C.c_int64), ("num_trajectory_samples", C.c_int64), ("num_elite", C.c_int64),
                ("iter_max", C.c_int64), ("smoothing_factor", C.c_double), ("N", C.c_int64), ("m", C.c_int64),
                ("iter_current", C.c_int64), ("mu_init", _dp), ("Sigma_init", _dp), ("mu", _dp), ("Sigma", _dp)]


class NmSolver(C.Structure):
    _fields_ = [("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double), ("eps", C.c_double), ("lam", C.c_double),
                ("iter_max", C.c_int64), ("theta_high_init", C.c_double), ("theta_low_init", C.c_double),
                ("iter_current", C.c_int64), ("theta_high", C.c_double), ("theta_low", C.c_double),
                ("has_c_high", C.c_int32), ("has_c_low", C.c_int32), ("c_high", C.c_double), ("c_low", C.c_double),
                ("n_solves", C.c_int64), ("n_batches", C.c_int64)]


EXPORTS = [
    "rat_version", "rat_last_error", "rat_default_ileqg_opts", "rat_create", "rat_destroy", "rat_set_ileqg_opts",
    "rat_problem_set", "rat_ileqg_solve_batch", "rat_set_initial", "rat_ileqg_solve_batch_dev", "rat_ileqg_solve",
    "rat_rollout_open", "rat_rollout_feedback", "rat_rollout_noisy", "rat_integrate_cost", "rat_approximate_model", "rat_dp_gain_sweep",
    "rat_dp_policy_eval", "rat_ce_default", "rat_ce_initialize", "rat_ce_set_stream", "rat_ce_seed",
    "rat_ce_stream_pos", "rat_ce_get_positive_samples", "rat_ce_compute_cost", "rat_ce_compute_cost_dev", "rat_ce_compute_cost_enqueue", "rat_ce_begin_step", "rat_ce_draw",
    "rat_ce_update", "rat_ce_draw_stream", "rat_ce_step", "rat_ce_solve", "rat_nm_default", "rat_nm_initialize",
    "rat_nm_compute_cost", "rat_nm_step", "rat_nm_solve", "rat_pets_problem_set", "rat_pets_initialize",
    "rat_pets_compute_cost", "rat_pets_sample_controls", "rat_pets_update", "rat_pets_step", "rat_pets_solve", "rat_profile_enable", "rat_profile_reset", "rat_profile_get",
    "rat_stream", "rat_layout_info",
    "rat_dp_gain_sweep_batch", "rat_dp_policy_eval_batch",
    "rat_shard_bounds", "rat_create_multi", "rat_multi_destroy", "rat_multi_n_devices", "rat_multi_handle", "rat_multi_uses_rccl",
    "rat_multi_allgathers", "rat_multi_problem_set", "rat_multi_set_initial", "rat_multi_ce_compute_cost", "rat_multi_ce_step",
    "rat_multi_ce_solve", "rat_multi_pets_problem_set", "rat_multi_pets_compute_cost",
    "rat_multi_ce_compute_cost_ex", "rat_multi_ileqg_solve_batch", "rat_multi_is_logical", "rat_set_path", "rat_get_path", "rat_ce_compute_cost_enqueue_ex",
    "rat_debug_set", "rat_debug_get", "rat_ce_update_dev",
]

_lib = None


def lib():
    """Load the HIP library; raise loudly if it is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RatError(f"{SO_PATH} is missing: build it with `make -C {os.path.dirname(SO_PATH)}` "
                           "(there is no CPU fallback)")
        _lib = C.CDLL(SO_PATH)
        _lib.rat_last_error.restype = C.c_char_p
        _lib.rat_stream.restype = C.c_void_p
        _lib.rat_ce_stream_pos.restype = C.c_int64
        _lib.rat_stream.argtypes = [C.c_void_p]
        _lib.rat_destroy.argtypes = [C.c_void_p]
        _lib.rat_destroy.restype = None
        _lib.rat_multi_destroy.argtypes = [C.c_void_p]
        _lib.rat_multi_destroy.restype = None
        _lib.rat_multi_handle.argtypes = [C.c_void_p, C.c_int32]
        _lib.rat_multi_handle.restype = C.c_void_p
        _lib.rat_multi_allgathers.argtypes = [C.c_void_p]
        _lib.rat_multi_allgathers.restype = C.c_int64
        _lib.rat_set_path.argtypes = [C.c_void_p, C.c_int32]
        _lib.rat_get_path.argtypes = [C.c_void_p, C.c_int64]
        _lib.rat_get_path.restype = C.c_int32
        _lib.rat_debug_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        _lib.rat_debug_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    return _lib


def check(rc):
    if rc != 0:
        raise RatError(f"{RC_NAMES.get(rc, rc)}: {lib().rat_last_error().decode()}")


def P(a):
    return None if a is None else a.ctypes.data_as(_dp)


def PI(a):
    return None if a is None else a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def cm3(a):
    """(T, rows, cols) -> flat buffer, time slowest, column-major inside (Vector{Matrix} of Julia)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).transpose(0, 2, 1)).ravel()


def from_cm3(buf, T, rows, cols):
    return np.asarray(buf, dtype=np.float64).reshape(T, cols, rows).transpose(0, 2, 1).copy()


def make_desc(prob):
    t = prob.c_tables()
    keep = {k: f64(v) for k, v in t.items() if isinstance(v, np.ndarray)}
    d = ProblemDesc()
    for k in ("model", "n", "m", "N", "cost_tv", "W_tv"):
        setattr(d, k, int(t[k]))
    for k in ("q0f", "kappa", "pl_a", "pl_b", "pl_p", "pl_pu", "pl_cx", "pl_cu", "pl_h"):
        setattr(d, k, float(t[k]))
    for k, v in keep.items():
        setattr(d, k, P(v))
    return d, keep
