"""Multi-GPU sharding of the Cross-Entropy cost evaluation: one process per GPU, theta-samples
split in contiguous blocks, one all-gather of per-sample costs (RCCL over xGMI when the process group
is ``nccl``; ``gloo`` on CPU for tests).

Replaces the reference's ``@sync/@async remotecall_fetch`` fan-out over Julia worker processes
(cross_entropy_bilevel_optimization.jl:180-192): assignment there is round-robin ``2 + mod(i, nprocs-1)``
(:181); assignment does not affect results, so contiguous blocks are used (one coalesced gather).
The message is B/G doubles per rank (1 KiB at B = 1024, G = 8): latency-bound, so it is a single
collective per CE batch and nothing else crosses ranks.  Every rank draws the same theta array (same
N(0,1) stream), so the elite selection of ``rat_ce_update`` is replicated instead of broadcast.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable

import numpy as np
import torch
import torch.distributed as dist

from . import _native as nv


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank; blocks differ by at most one sample."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_values(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """All-gather ragged contiguous shards of a length-B vector (pads to ceil(B/G) per rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local.clone()
    chunk = -(-B // world)
    buf = torch.full((chunk,), float("nan"), dtype=local.dtype, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty((world * chunk,), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group) if local.is_cuda else dist.all_gather(
        list(out.view(world, chunk).unbind(0)), buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        parts.append(out[r * chunk: r * chunk + (hi - lo)])
    return torch.cat(parts)


def compute_cost_sharded(theta: np.ndarray, kl_bound: float, evaluate_shard: Callable[[torch.Tensor], torch.Tensor],
                         device="cpu", group=None, assignment="contiguous") -> np.ndarray:
    """compute_cost (:173-195) with the value fan-out sharded over the ranks of ``group``.

    ``evaluate_shard(theta_shard)`` returns the iLEQG values (Inf on failure) of this rank's block as a tensor
    on ``device``; product code passes Context.solve_batch_dev wrapped by ``gpu_evaluator``.
    ``assignment``: "contiguous" blocks, or "interleaved" -- the samples sorted by theta are dealt round-robin (rank r takes every
    world-th one), the reference's own pattern (:181).  Iteration counts grow with theta on nonlinear problems and infeasible samples
    stop at initialize!, so dealing the sorted samples evens out the work per rank; results do not depend on the assignment."""
    B = int(theta.size)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if assignment == "interleaved" and world > 1:
        order = np.argsort(theta, kind="stable")
        mine = order[rank::world]
        th = torch.as_tensor(np.ascontiguousarray(theta[mine]), dtype=torch.float64, device=device)
        val = evaluate_shard(th)
        chunk = -(-B // world)
        buf = torch.full((chunk,), float("nan"), dtype=torch.float64, device=val.device)
        buf[: val.numel()] = val
        out = torch.empty((world * chunk,), dtype=torch.float64, device=val.device)
        if val.is_cuda:
            dist.all_gather_into_tensor(out, buf, group=group)
        else:
            dist.all_gather(list(out.view(world, chunk).unbind(0)), buf, group=group)
        out = out.cpu().numpy().reshape(world, chunk)
        allv = np.empty(B)
        for r in range(world):
            idx = order[r::world]
            allv[idx] = out[r, : idx.size]
        return allv + kl_bound / theta                                              # :193
    assert assignment in ("contiguous", "interleaved")
    lo, hi = shard_bounds(B, world, rank)
    th = torch.as_tensor(np.ascontiguousarray(theta[lo:hi]), dtype=torch.float64, device=device)
    val = evaluate_shard(th)
    allv = allgather_values(val, B, group)
    return allv.cpu().numpy() + kl_bound / theta                                    # :193


def gpu_evaluator(ctx):
    """evaluate_shard for a ratilqr Context: theta and values stay in HBM (device-pointer ABI)."""

    def ev(th: torch.Tensor) -> torch.Tensor:
        assert th.is_cuda and th.dtype == torch.float64
        out = torch.empty_like(th)
        if th.numel():
            torch.cuda.current_stream().synchronize()
            ctx.solve_batch_dev(th.data_ptr(), th.numel(), out.data_ptr())
        return out

    return ev


def step_sharded(ce_solver, x_unused, kl_bound, z: np.ndarray, zpos: int, evaluate_shard, device="cpu", group=None):
    """step! (:252-335) with sharded cost evaluation.  Returns (theta, cost, new zpos)."""
    L = nv.lib()
    c = ce_solver.c
    nv.check(L.rat_ce_begin_step(C.byref(c)))
    B = int(c.num_samples)
    theta = np.zeros(B)
    zp = C.c_int64(zpos)
    z = nv.f64(z)
    for redraws in range(1001):
        nv.check(L.rat_ce_draw_stream(C.byref(c), nv.P(z), C.c_int64(z.size), C.byref(zp), nv.P(theta)))
        cost = compute_cost_sharded(theta, kl_bound, evaluate_shard, device=device, group=group)
        c.n_solves += B
        redraw = C.c_int32()
        nv.check(L.rat_ce_update(C.byref(c), nv.P(theta), nv.P(nv.f64(cost)), C.byref(redraw)))
        if not redraw.value:
            return theta, cost, zp.value
        c.n_redraws += 1
    raise nv.RatError("CE redraw loop cut after 1000 redraws (reference would spin, SURVEY App. B.11)")


def solve_sharded(ce_solver, kl_bound, z: np.ndarray, evaluate_shard, final_solve, device="cpu", group=None):
    """solve! (:364-415): CE iterations with sharded costs, then the final solve at theta_opt on every rank
    (``final_solve(theta) -> (ok, x, l, L, value)``).  Returns (theta_opt, x, l, L, value, theta_min, theta_max)."""
    assert kl_bound >= 0, "KL Divergence Bound must be non-negative"
    L = nv.lib()
    c = ce_solver.c
    L.rat_ce_initialize(C.byref(c))
    zpos = 0
    tmin = tmax = 0.0
    if kl_bound > 0:
        while c.iter_current < c.iter_max:
            _, _, zpos = step_sharded(ce_solver, None, kl_bound, z, zpos, evaluate_shard, device=device, group=group)
        tmin, tmax = c.theta_min, c.theta_max
        theta_opt = tmax if c.use_theta_max else c.mu
    else:
        theta_opt = 0.0
    for _ in range(10001):
        ok, x, l, Lg, value = final_solve(theta_opt)
        if ok:
            if kl_bound > 0:
                return theta_opt, x, l, Lg, value + kl_bound / theta_opt, tmin, tmax
            return theta_opt, x, l, Lg, value, 0.0, 0.0
        theta_opt = max(0.0, theta_opt - c.sigma)                                     # :412
    raise nv.RatError("final-solve retry loop cut (reference would spin, SURVEY App. B.15)")


# ---- PETS (pets.jl:100-126): control samples sharded over the ranks -----------------------------------------------------------
def pets_compute_cost_sharded(control_sequence_array: np.ndarray, evaluate_shard, device="cpu", group=None) -> np.ndarray:
    """compute_cost of the PETS solver (pets.jl:100-126) with the S control samples split in contiguous blocks over the ranks.

    All K noisy trajectories of a control sample stay on one rank (their mean is the sample's cost, :150), so the only exchange
    is one all-gather of S/G doubles per rank.  ``evaluate_shard(controls_block, lo)`` returns the block's costs as a tensor
    on ``device`` (``lo`` = global index of the block's first sample: noise streams are addressed by global sample index, which
    makes the result independent of the number of ranks).  Elite selection / refit run replicated on every rank."""
    ctrl = np.ascontiguousarray(control_sequence_array, dtype=np.float64)
    S = ctrl.shape[0]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(S, world, rank)
    local = evaluate_shard(ctrl[lo:hi], lo)
    return allgather_values(torch.as_tensor(local, dtype=torch.float64, device=device), S, group).cpu().numpy()


def pets_gpu_evaluator(direct_solver, problem, x, use_true_model=False, streams=None, seed=0):
    """evaluate_shard for pets_compute_cost_sharded on this rank's GPU.  ``streams=(zn, zu)`` are the full injected draw arrays
    of compute_cost_serial (addressed by global sample index); without them the device generator is keyed by (seed, block)."""
    ctx = direct_solver.context(problem)
    K, N, n = int(direct_solver.c.num_trajectory_samples), problem.N, problem.n
    xx = nv.f64(x)

    def ev(block: np.ndarray, lo: int) -> torch.Tensor:
        Sb = block.shape[0]
        cost = np.zeros(Sb)
        if Sb:
            zn = zu = None
            if streams is not None:
                zn = nv.f64(np.asarray(streams[0])[lo * K * N * n: (lo + Sb) * K * N * n])
                zu = None if streams[1] is None else nv.f64(np.asarray(streams[1])[lo * K * N: (lo + Sb) * K * N])
            nv.check(nv.lib().rat_pets_compute_cost(ctx.h, nv.P(xx), nv.P(nv.f64(block)), C.c_int64(Sb), C.c_int64(K), int(use_true_model),
                                                    nv.P(zn), nv.P(zu), C.c_uint64(int(seed) + 0x9E3779B9 * int(lo)), nv.P(cost)))
        return torch.as_tensor(cost, dtype=torch.float64)

    return ev
