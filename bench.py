#!/usr/bin/env python3
"""bench.py -- iLEQG solves/sec (N=50, n=12, m=4) at CE batch = 1024 per GPU.

A "step" is one compute_cost-equivalent pass (cross_entropy_bilevel_optimization.jl:173-195) over a CE batch of
1024 theta-samples on every rank: 1024 complete iLEQG solves (initialize! + iterations with line search) on the
synthetic LQ-plus-noise problem of SURVEY.md section 8(d), followed -- when N > 1 -- by the all-gather of the
per-sample costs (RCCL).  theta, x0, u0 and the problem tables are resident in HBM before the timed region.
Weak scaling: every rank owns its own 1024-sample CE batch; value = N * 1024 * K / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--spec-eps E] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def algo_bytes(n=12, m=4, N=50):
    """SURVEY.md section 8(d) "algorithmic bytes per unit" (unfused three-kernel formulation), in bytes.
    tile bundle = N (2n^2 + 2nm + m^2 + n + m + 1) + n^2 + n + 1 doubles; x = (N+1) n; l, dl = N m; L = N m n."""
    tile = N * (2 * n * n + 2 * n * m + m * m + n + m + 1) + n * n + n + 1
    x, l, L = (N + 1) * n, N * m, N * m * n
    d = {
        "rollout_candidate": 8 * ((x + l + l + L) + (x + l)),      # reads xbar, l, dl, L; writes x, u
        "rollout_init": 8 * (l + x),                                # open loop: reads u, writes x
        "linearise": 8 * ((x + l) + tile),
        "sweep_eval": 8 * (tile + L + 1),
        "sweep_gain": 8 * (tile + L + l + 1),
    }
    d["candidate"] = d["rollout_candidate"] + d["linearise"] + d["sweep_eval"]           # 395.6 KB
    d["init"] = d["rollout_init"] + d["linearise"] + d["sweep_eval"]                     # 368.3 KB
    return d


def algo_bytes_per_candidate(prob=None):
    return algo_bytes()["sweep_eval"]


def algo_bytes_of_solves(iters, ls_evals):
    """Bytes of complete solves: initialize! + one gain sweep per step! + one candidate per line-search evaluation
    (1.54 MB for the 2-iteration, 2-evaluation LQ solve of SURVEY.md section 8d)."""
    a = algo_bytes()
    return float(len(iters)) * a["init"] + float(np.sum(iters)) * a["sweep_gain"] + float(np.sum(ls_evals)) * a["candidate"]


FP64_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak (they are one datapath, profiles/r01_ubench_fp64_pipe.md)


def algo_flops_of_solves(iters, ls_evals, N=50):
    """SURVEY.md section 8(d) "algorithmic flops per unit" (factored form, App. D): 22.2 kflop per backward step, rollout 24 kflop and
    linearise 20 kflop per trajectory; a solve runs 1 + iters + ls_evals sweeps and 1 + ls_evals rollouts / linearisations
    (5.7 Mflop for the 2-iteration, 2-evaluation LQ solve)."""
    sweeps = float(len(iters)) + float(np.sum(iters)) + float(np.sum(ls_evals))
    trajs = float(len(iters)) + float(np.sum(ls_evals))
    return sweeps * 22.2e3 * N + trajs * (24e3 + 20e3)


def draw_theta(B, seed):
    """Positive samples of N(1, 2) -- the CE solver's first-iteration distribution (mu_init=1, sigma_init=2)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < B:
        z = 1.0 + 2.0 * rng.standard_normal(B)
        out.extend(z[z > 0.0].tolist())
    return np.array(out[:B])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--spec-eps", type=int, default=1, help="E speculative line-search step sizes per sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-second", action="store_true", help="skip the secondary E=8 measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_distributed.py runs two ranks on ONE GPU): the collective backend, and all ranks on device 0
    backend = os.environ.get("RATILQR_BENCH_BACKEND", "nccl")
    if os.environ.get("RATILQR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    # (test hook: run the collective path even with one rank -- RCCL on the handle's stream on a single-GPU box)
    multi = world > 1 or os.environ.get("RATILQR_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # RCCL
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import ratilqr.jl_amd as rat

    B, E, K, W = args.batch, args.spec_eps, args.steps, args.warmup
    prob, x0, u0 = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=0)
    ctx = rat.Context(prob, max_batch=B, spec_eps=E, device=local_rank)
    ctx.set_initial(x0, u0)
    kl_bound = 0.1
    theta_h = draw_theta(B, seed=1000 + rank)
    theta = torch.as_tensor(theta_h, dtype=torch.float64, device=dev)
    value = torch.empty(B, dtype=torch.float64, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    ls = torch.empty(B, dtype=torch.int32, device=dev)
    cost = torch.empty(B, dtype=torch.float64, device=dev)
    cost_all = torch.empty(world * B, dtype=torch.float64, device=dev) if multi else cost
    torch.cuda.synchronize()

    # the handle's own HIP stream, seen by torch: the batch, the cost all-gather and the next batch are ordered on it on the device,
    # with no host round trip between steps (the timed region ends with a device synchronisation)
    hstream = torch.cuda.ExternalStream(ctx.stream, device=dev)

    def step():
        # compute_cost (cross_entropy...jl:173-195): B complete solves and cost = value + kl/theta (:193), all on the device
        # (one kernel launch on the fused path), then the per-sample costs go to every rank
        ctx.compute_cost_enqueue(theta.data_ptr(), B, kl_bound, cost.data_ptr())
        if multi and backend == "nccl":
            with torch.cuda.stream(hstream):                  # RCCL waits for the batch and the next batch waits for RCCL
                dist.all_gather_into_tensor(cost_all, cost)
        elif multi:                                           # host-staged collective (test hook)
            hstream.synchronize()
            parts = [torch.empty(B, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, cost.cpu())
            cost_all.copy_(torch.cat(parts))

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    # untimed pass with HIP events around every kernel kind: per-kernel breakdown (events perturb the stream, so the
    # timed region below only brackets the dominant kernel, whose duration feeds the roofline object)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(max(2, min(K, 5))):
        step()
    torch.cuda.synchronize()
    prof_all = ctx.profile_get()
    n_all = max(2, min(K, 5))
    fused = prof_all["solve_fused"]["launches"] > 0
    main_kind = "solve_fused" if fused else "sweep_eval"
    ctx.profile(True, kinds=[main_kind])
    ctx.profile_reset()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile(False)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # untimed: the per-sample outputs of the same batch (statuses, iteration and line-search counts for the report; the costs of
    # the timed path are checked against them)
    ctx.solve_batch_dev(theta.data_ptr(), B, value.data_ptr(), status.data_ptr(), iters.data_ptr(), ls.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(cost.cpu().numpy(), value.cpu().numpy() + kl_bound / theta_h), "compute_cost_dev disagrees with value + kl/theta"
    st_h, it_h, ls_h = status.cpu().numpy(), iters.cpu().numpy(), ls.cpu().numpy()
    feasible = float(np.mean((st_h == 0) | (st_h == 3)))

    # secondary measurement in the same run: E = 8 speculative step sizes per sample (BASELINE config 3's "x 8 line-search
    # eps"); identical results, 8x the candidate work on this problem (every first candidate is accepted).
    second = None
    if world == 1 and E != 8 and not args.no_second:
        ctx8 = rat.Context(prob, max_batch=B, spec_eps=8, device=local_rank)
        ctx8.set_initial(x0, u0)
        v8 = torch.empty(B, dtype=torch.float64, device=dev)
        for _ in range(6):                  # (as many untimed batches as the primary measurement has behind it when its timing starts)
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        ctx8.profile(True, kinds=["sweep_eval"])
        ctx8.profile_reset()
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        K8 = max(3, K // 3)
        for _ in range(K8):
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        torch.cuda.synchronize()
        e8 = time.perf_counter() - t8
        p8 = ctx8.profile_get()["sweep_eval"]
        bpt = algo_bytes_per_candidate(prob)
        ach8 = bpt * (p8["trajectories"] / max(p8["launches"], 1)) / (p8["ms"] / max(p8["launches"], 1) * 1e-3) / 1e9
        second = {"spec_eps": 8, "value": B * K8 / e8, "unit": "solves/s", "ms_per_step": e8 / K8 * 1e3, "steps": K8,
                  "values_identical_to_primary": bool(torch.equal(v8, value)),
                  "roofline": {"bound": "hbm", "achieved": ach8, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach8 / HBM_PEAK_GBS,
                               "trajectories_per_launch": p8["trajectories"] / max(p8["launches"], 1),
                               "avg_launch_ms": p8["ms"] / max(p8["launches"], 1)}}
        del ctx8

    # second secondary: a nonlinear workload (cubic drift kappa x^3 on the same tables): more iterations per solve and
    # theta-dependent iteration counts, i.e. divergent per-sample control flow inside one launch
    nonlin = None
    if world == 1 and not args.no_second:
        probn, x0n, u0n = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=0, kappa=0.05)
        ctxn = rat.Context(probn, max_batch=B, spec_eps=E, device=local_rank)
        ctxn.set_initial(x0n, u0n)
        vn = torch.empty(B, dtype=torch.float64, device=dev)
        stn = torch.empty(B, dtype=torch.int32, device=dev)
        itn = torch.empty(B, dtype=torch.int32, device=dev)
        lsn = torch.empty(B, dtype=torch.int32, device=dev)
        for _ in range(3):
            ctxn.solve_batch_dev(theta.data_ptr(), B, vn.data_ptr(), stn.data_ptr(), itn.data_ptr(), lsn.data_ptr())
        torch.cuda.synchronize()
        tn = time.perf_counter()
        Kn = max(3, K // 3)
        for _ in range(Kn):
            ctxn.solve_batch_dev(theta.data_ptr(), B, vn.data_ptr(), stn.data_ptr(), itn.data_ptr(), lsn.data_ptr())
        torch.cuda.synchronize()
        en = time.perf_counter() - tn
        sn = stn.cpu().numpy()
        nonlin = {"workload": "the same problem with cubic drift kappa = 0.05 (f = A x + B u + kappa x^3)", "value": B * Kn / en,
                  "unit": "solves/s", "ms_per_step": en / Kn * 1e3, "steps": Kn,
                  "feasible_fraction": float(np.mean((sn == 0) | (sn == 3))), "mean_iters": float(itn.float().mean().item()),
                  "mean_ls_evals": float(lsn.float().mean().item()),
                  "algorithmic_GBps": algo_bytes_of_solves(itn.cpu().numpy(), lsn.cpu().numpy()) * Kn / en / 1e9}
        del ctxn

    # per-phase breakdown of the same batch on the round-based path (one launch per phase; what the fused kernel replaces)
    unfused = None
    if fused and world == 1:
        os.environ["RATILQR_FUSED"] = "0"
        ctxu = rat.Context(prob, max_batch=B, spec_eps=E, device=local_rank)
        del os.environ["RATILQR_FUSED"]
        ctxu.set_initial(x0, u0)
        vu = torch.empty(B, dtype=torch.float64, device=dev)
        for _ in range(2):
            ctxu.solve_batch_dev(theta.data_ptr(), B, vu.data_ptr())
        ctxu.profile(True)
        ctxu.profile_reset()
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for _ in range(5):
            ctxu.solve_batch_dev(theta.data_ptr(), B, vu.data_ptr())
        torch.cuda.synchronize()
        eu = (time.perf_counter() - tu) / 5
        pu = ctxu.profile_get()
        pe = pu["sweep_eval"]
        ach_e = algo_bytes()["sweep_eval"] * (pe["trajectories"] / max(pe["launches"], 1)) / (pe["ms"] / max(pe["launches"], 1) * 1e-3) / 1e9
        unfused = {"ms_per_step_with_events": eu * 1e3, "values_identical_to_fused": bool(torch.equal(vu, value)),
                   "kernel_ms_per_step": {k: v["ms"] / 5 for k, v in pu.items() if v["launches"]},
                   "sweep_eval_roofline": {"achieved": ach_e, "unit": "GB/s", "frac": ach_e / HBM_PEAK_GBS,
                                           "avg_launch_ms": pe["ms"] / max(pe["launches"], 1)}}
        del ctxu

    if rank == 0:
        lay = ctx.layout_info()
        # Algorithmic bytes (information content, not the padded HBM records).
        #  fused path: the dominant kernel is the whole solve -- initialize! + iters gain sweeps + ls_evals candidates
        #              per sample (1.54 MB for this workload's 2-iteration solves), summed over the launch's samples;
        #  round-based path: the policy-evaluation sweep, 187.3 KB per candidate.
        pe = prof[main_kind]
        avg_ms = pe["ms"] / max(pe["launches"], 1)
        traj_per_launch = pe["trajectories"] / max(pe["launches"], 1)
        if fused:
            bytes_per_launch = algo_bytes_of_solves(it_h, ls_h)
            bytes_per_traj = bytes_per_launch / B
            kernel_name = "solve_fused_kernel (one persistent wavefront per theta-sample: whole solve!)"
        else:
            bytes_per_traj = algo_bytes_per_candidate(prob)
            bytes_per_launch = bytes_per_traj * traj_per_launch
            kernel_name = "sweep_kernel<eval> (policy-evaluation Riccati sweep of line-search candidates)"
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"{main_kind}_E{E}_B{B}", None)
            except Exception:
                traffic = None
        out = {
            "metric": "iLEQG solves/sec (N=50, n=12, m=4) at CE batch=1024",
            "value": world * B * K / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "batched iLEQG solves of one CE batch (compute_cost): synthetic LQ-plus-noise, N=50, n=12, m=4, "
                            "W=1e-3 I, theta ~ N(1,2)>0, kl=0.1, iLEQG defaults",
                "ce_batch_per_gpu": B, "global_batch": world * B, "spec_eps": E,
                "parallelism": f"theta-shards x{world}, cost all-gather" if world > 1 else "single GPU",
                "feasible_fraction": feasible, "mean_iters": float(it_h.mean()), "mean_ls_evals": float(ls_h.mean()),
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "bytes_per_trajectory": bytes_per_traj, "hbm_record_bytes_per_trajectory": lay["tile_bytes"] + lay["L_bytes"] + 8,
                "trajectories_per_launch": traj_per_launch,
                "avg_launch_ms": avg_ms, "launches": pe["launches"],
            },
        }
        if fused:           # the sweep sits at the fp64 ridge -- report the FP64 fraction of the same launches beside the HBM one
            fl = algo_flops_of_solves(it_h, ls_h)
            out["roofline"]["fp64"] = {"achieved": fl / (avg_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": fl / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "flops_per_solve": fl / B}
        out.update({
            "kernel_ms_per_step": {k: v["ms"] / n_all for k, v in prof_all.items()},
        })
        if unfused is not None:
            out["round_based_path"] = unfused
        if second is not None:
            out["secondary_spec_eps8"] = second
        if nonlin is not None:
            out["secondary_nonlinear"] = nonlin
        if world == 1 and not args.no_cpu:
            from oracle import oracle as orc

            P = orc.Problem(prob)
            cores = os.cpu_count() or 1
            n_done, t_cpu = 0, 0.0
            chunk = max(64, 32 * cores)
            while t_cpu < args.cpu_seconds:
                th = draw_theta(chunk, seed=77 + n_done)
                t1 = time.perf_counter()
                orc.compute_value_batch(P, x0, u0, th, nthreads=cores)
                t_cpu += time.perf_counter() - t1
                n_done += chunk
            import shutil
            jl = shutil.which("julia")                          # probed at run time: the reference itself is never on the GPU box
            julia_note = ("the Julia reference is not runnable (no julia binary)" if jl is None else
                          f"a julia binary exists ({jl}) but the reference package and its dependencies do not travel to this box")
            th1 = draw_theta(32, seed=55)                       # the same oracle on one thread
            t1 = time.perf_counter()
            orc.compute_value_batch(P, x0, u0, th1, nthreads=1)
            t_one = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": n_done / t_cpu, "unit": "solves/s", "cores": cores, "kind": "port", "value_1thread": 32 / t_one,
                "sample": f"{n_done} solves of the same workload (theta ~ N(1,2)>0) by the C oracle, OpenMP one sample per "
                          f"thread on {cores} threads, {t_cpu:.1f} s; " + julia_note,
            }
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
