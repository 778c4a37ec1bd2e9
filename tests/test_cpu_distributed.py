"""world_size-2 gloo test of the theta-sharding layer (ratilqr.jl_amd.distributed): contiguous shards,
one all-gather of per-sample costs, replicated elite selection.  The cost evaluator is injected: on the
GPU box it is Context.solve_batch_dev (tests/test_gpu_distributed.py); here (CPU, no GPU) the oracle stands
in as the evaluator so that the sharding / gather / CE bookkeeping logic itself is what is under test."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ratilqr.jl_amd as rat
from ratilqr.jl_amd import distributed as rd
from oracle import oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    return rat.synthetic_lq_problem(n=4, m=2, N=20, seed=1)


def _worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob, x0, u = _problem()
    P = orc.Problem(prob)
    calls = []

    def evaluate(th):
        calls.append(th.numel())
        v, _, _, _ = orc.compute_value_batch(P, x0, u, th.numpy(), nthreads=1)
        return torch.as_tensor(v, dtype=torch.float64)

    def final(theta):
        s = orc.ILEQGSolver(P)
        rc = s.solve(x0, u, theta)
        return rc in (0, 3), s.x_array, s.l_array, s.L_array, s.s.value_current

    z = np.random.default_rng(99).standard_normal(20000)
    solver = rat.CrossEntropyBilevelOptimizationSolver(num_samples=B, num_elite=4)
    res = rd.solve_sharded(solver, 0.1, z, evaluate, final)
    lo, hi = rd.shard_bounds(B, world, rank)
    assert all(c == hi - lo for c in calls)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([res[0], res[4], res[5], res[6], solver.c.mu, solver.c.sigma,
                                                             solver.c.mu_init, solver.c.n_solves]))
    dist.destroy_process_group()


def _run(world, B, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), f"r{r}.npy")) for r in range(world)]


def test_shard_bounds_cover_everything():
    for B in (1, 7, 13, 1024):
        for G in (1, 2, 3, 8):
            spans = [rd.shard_bounds(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_ce_matches_single_process_oracle(tmp_path):
    B = 13                                   # ragged: 7 + 6
    got = _run(2, B, tmp_path)
    assert np.array_equal(got[0], got[1])    # replicated decisions: every rank ends with identical state
    prob, x0, u = _problem()
    z = np.random.default_rng(99).standard_normal(20000)
    oc = orc.CrossEntropyBilevelOptimizationSolver(z, num_samples=B, num_elite=4)
    rc, th, x, l, L, val, tmin, tmax = oc.solve(orc.Problem(prob), x0, u, 0.1)
    assert rc == 0
    ref = np.array([th, val, tmin, tmax, oc.c.mu, oc.c.sigma, oc.c.mu_init, oc.c.n_solves])
    assert np.array_equal(got[0], ref)       # same evaluator, same stream -> bitwise the same CE trajectory


# ---- PETS: control samples sharded over the ranks (BASELINE config 5) ------------------------------------------------------------
def _pets_setup():
    prob = rat.LQGenerativeProblem(np.eye(2), np.eye(2), 6, ("uniform", 0.0, 1.0), l1u=1.0, q0f=1.0)
    rng = np.random.default_rng(7)
    S, K = 11, 5                                                    # ragged over 2 ranks: 6 + 5
    ctrl = rng.random((S, prob.N, 2))
    zn = rng.random(S * K * prob.N * 2)
    return prob, ctrl, zn, S, K


def _pets_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob, ctrl, zn, S, K = _pets_setup()
    G = orc.GenProblem(prob)
    seen = []

    def evaluate(block, lo):                                        # the oracle stands in for the device rollouts
        seen.append((lo, block.shape[0]))
        z = zn[lo * K * prob.N * 2: (lo + block.shape[0]) * K * prob.N * 2]
        return torch.as_tensor(orc.pets_compute_cost(G, np.zeros(2), block, K, False, z), dtype=torch.float64)

    cost = rd.pets_compute_cost_sharded(ctrl, evaluate)
    assert seen == [rd.shard_bounds(S, world, rank)[0:1] + (rd.shard_bounds(S, world, rank)[1] - rd.shard_bounds(S, world, rank)[0],)]
    np.save(os.path.join(out_dir, f"p{rank}.npy"), cost)
    dist.destroy_process_group()


def test_two_rank_pets_cost_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_pets_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(str(tmp_path), f"p{r}.npy")) for r in range(2)]
    prob, ctrl, zn, S, K = _pets_setup()
    ref = orc.pets_compute_cost(orc.GenProblem(prob), np.zeros(2), ctrl, K, False, zn)
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], ref)


def _worker_interleaved(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob, x0, u = _problem()
    P = orc.Problem(prob)
    seen = []

    def evaluate(th):
        seen.append(th.numpy().copy())
        v, _, _, _ = orc.compute_value_batch(P, x0, u, th.numpy(), nthreads=1)
        return torch.as_tensor(v, dtype=torch.float64)

    theta = np.abs(1.0 + 2.0 * np.random.default_rng(5).standard_normal(13)) + 0.01      # ragged: 13 samples on 3 ranks
    theta[4] = 500.0                                                                      # infeasible -> Inf
    a = rd.compute_cost_sharded(theta, 0.1, evaluate, assignment="interleaved")
    b = rd.compute_cost_sharded(theta, 0.1, evaluate, assignment="contiguous")
    assert np.array_equal(a, b) and np.isposinf(a[4])
    assert np.array_equal(seen[0], np.sort(theta)[rank::world])                           # sorted samples dealt round-robin
    np.save(os.path.join(out_dir, f"i{rank}.npy"), a)
    dist.destroy_process_group()


def test_interleaved_assignment_gives_the_same_costs(tmp_path):
    port = _free_port()
    mp.spawn(_worker_interleaved, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    got = [np.load(os.path.join(str(tmp_path), f"i{r}.npy")) for r in range(3)]
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
    prob, x0, u = _problem()
    theta = np.abs(1.0 + 2.0 * np.random.default_rng(5).standard_normal(13)) + 0.01
    theta[4] = 500.0
    v, _, _, _ = orc.compute_value_batch(orc.Problem(prob), x0, u, theta, nthreads=1)
    assert np.array_equal(got[0], v + 0.1 / theta)
