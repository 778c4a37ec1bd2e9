"""Several GPUs behind one object, driven from one host thread: ``rat_create_multi`` and friends (include/ratilqr.h).

This is the in-library counterpart of ``distributed.py`` (one process per GPU over torch.distributed, what ``bench.py --gpus N`` and a
torchrun-style deployment use): a host that is ONE process -- the Julia package the north star names, a C program -- reaches all the
GPUs of a node through the C ABI alone; contiguous theta shards, one ncclAllGather of the per-sample costs per CE batch on the handles'
HIP streams, replicated elite selection (cross_entropy_bilevel_optimization.jl:180-192 is what it replaces)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nv


def shard_bounds(B: int, world: int, rank: int):
    """rat_shard_bounds (device-free): contiguous block [lo, hi) of `rank`."""
    lo, hi = C.c_int64(), C.c_int64()
    nv.check(nv.lib().rat_shard_bounds(C.c_int64(B), C.c_int32(world), C.c_int32(rank), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


class MultiContext:
    """One rat_multi bound to one problem: `devices` GPUs, a CE batch of up to `max_batch` samples split over them."""

    def __init__(self, problem, opts: nv.IleqgOpts | None = None, max_batch=1, spec_eps=1, devices=(0,)):
        L = nv.lib()
        self.problem = problem
        self.n, self.m, self.N = problem.n, problem.m, problem.N
        self.devices = [int(d) for d in devices]
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        self.m_ = C.c_void_p()
        nv.check(L.rat_create_multi(C.byref(opts) if opts is not None else None, int(max_batch), int(spec_eps), len(self.devices), dev,
                                    C.byref(self.m_)))
        import weakref
        self._fin = weakref.finalize(self, L.rat_multi_destroy, self.m_)
        desc, self._keep = nv.make_desc(problem)
        nv.check(L.rat_multi_problem_set(self.m_, C.byref(desc)))

    n_devices = property(lambda s: int(nv.lib().rat_multi_n_devices(s.m_)))
    uses_rccl = property(lambda s: bool(nv.lib().rat_multi_uses_rccl(s.m_)))
    allgathers = property(lambda s: int(nv.lib().rat_multi_allgathers(s.m_)))

    def handle(self, i=0):
        return C.c_void_p(nv.lib().rat_multi_handle(self.m_, int(i)))

    def set_stream(self, z):
        """Standard-normal stream of the CE draws (consumed through device 0's handle)."""
        self._z = nv.f64(z)
        nv.check(nv.lib().rat_ce_set_stream(self.handle(0), nv.P(self._z), C.c_int64(self._z.size)))

    def compute_cost(self, x0, u, theta, kl_bound):                       # compute_cost  :173-195
        theta = nv.f64(theta)
        cost = np.zeros(theta.size)
        nv.check(nv.lib().rat_multi_ce_compute_cost(self.m_, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), nv.P(theta), C.c_int64(theta.size),
                                                    C.c_double(kl_bound), nv.P(cost)))
        return cost

    is_logical = property(lambda s: bool(nv.lib().rat_multi_is_logical(s.m_)))

    def compute_cost_ex(self, x0, u, theta, kl_bound):
        """compute_cost with the gathered per-sample statistics of every shard: (cost, status, iters, ls_evals)."""
        theta = nv.f64(theta)
        B = theta.size
        cost, st, it, ls = np.zeros(B), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        nv.check(nv.lib().rat_multi_ce_compute_cost_ex(self.m_, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), nv.P(theta), C.c_int64(B),
                                                       C.c_double(kl_bound), nv.P(cost), nv.PI(st), nv.PI(it), nv.PI(ls)))
        return cost, st, it, ls

    def solve_batch(self, x0, u, theta):
        """rat_ileqg_solve_batch over all devices: (value, status, iters, ls_evals)."""
        theta = nv.f64(theta)
        B = theta.size
        val, st, it, ls = np.zeros(B), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        nv.check(nv.lib().rat_multi_ileqg_solve_batch(self.m_, nv.P(nv.f64(x0)), nv.P(nv.f64(u)), nv.P(theta), C.c_int64(B),
                                                      nv.P(val), nv.PI(st), nv.PI(it), nv.PI(ls)))
        return val, st, it, ls

    def ce_step(self, c: nv.CeSolver, x0, u, kl_bound):                   # step!  :252-335
        B = int(c.num_samples)
        th, cost = np.zeros(B), np.zeros(B)
        nv.check(nv.lib().rat_multi_ce_step(self.m_, C.byref(c), nv.P(nv.f64(x0)), nv.P(nv.f64(u)), C.c_double(kl_bound), nv.P(th), nv.P(cost)))
        return th, cost

    def ce_solve(self, c: nv.CeSolver, x0, u, kl_bound):                  # solve!  :364-415
        n, m, N = self.n, self.m, self.N
        x, l, Lb = np.zeros((N + 1, n)), np.zeros((N, m)), np.zeros(m * n * N)
        th, val, tmin, tmax = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        nv.check(nv.lib().rat_multi_ce_solve(self.m_, C.byref(c), nv.P(nv.f64(x0)), nv.P(nv.f64(u)), C.c_double(kl_bound), C.byref(th), nv.P(x),
                                             nv.P(l), nv.P(Lb), C.byref(val), C.byref(tmin), C.byref(tmax)))
        return th.value, x, l, nv.from_cm3(Lb, N, m, n), val.value, tmin.value, tmax.value


class MultiPetsContext:
    """PETS cost evaluation (pets.jl:100-126) over several GPUs behind one object: rat_multi_pets_*."""

    def __init__(self, problem, devices=(0,)):
        L = nv.lib()
        self.problem = problem
        self.devices = [int(d) for d in devices]
        dev = (C.c_int32 * len(self.devices))(*self.devices)
        self.m_ = C.c_void_p()
        nv.check(L.rat_create_multi(None, 1, 1, len(self.devices), dev, C.byref(self.m_)))
        import weakref
        self._fin = weakref.finalize(self, L.rat_multi_destroy, self.m_)
        from .pets import make_gen_desc
        desc, self._keep = make_gen_desc(problem)
        nv.check(L.rat_multi_pets_problem_set(self.m_, C.byref(desc)))

    def compute_cost(self, x, control_sequence_array, K, use_true_model=False, streams=None, seed=0):
        ctrl = nv.f64(control_sequence_array)
        S = ctrl.shape[0]
        cost = np.zeros(S)
        zn = zu = None
        if streams is not None:
            zn = nv.f64(streams[0])
            zu = None if streams[1] is None else nv.f64(streams[1])
        nv.check(nv.lib().rat_multi_pets_compute_cost(self.m_, nv.P(nv.f64(x)), nv.P(ctrl), C.c_int64(S), C.c_int64(int(K)), int(bool(use_true_model)),
                                                      nv.P(zn), nv.P(zu), C.c_uint64(int(seed)), nv.P(cost)))
        return cost
