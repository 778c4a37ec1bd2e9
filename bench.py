#!/usr/bin/env python3
"""bench.py -- iLEQG solves/sec (N=50, n=12, m=4) at CE batch = 1024, on 1/2/4/8 MI355X.

A "step" is one compute_cost-equivalent pass (cross_entropy_bilevel_optimization.jl:173-195) over ONE Cross-Entropy batch of
1024 theta-samples: 1024 complete iLEQG solves (initialize! + iterations with line search) on the synthetic LQ-plus-noise
problem of SURVEY.md section 8(d).  theta, x0, u0 and the problem tables are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--spec-eps E] [--no-cpu] [--no-second]

N > 1 (one process per GPU over RCCL):
  * started by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE is set), or
  * started plainly as `python bench.py --gpus N`: this process then launches the N ranks itself -- BEFORE any GPU call -- and
    exits with their exit code; it fails (non-zero) when fewer than N devices are visible.
  `value` is the BASELINE.json configuration: the ONE CE batch of 1024 samples sharded over the N ranks in contiguous theta blocks
  (1024 / N solves per GPU) followed by the RCCL all-gather of the per-sample costs, i.e. STRONG scaling ("scaling": "strong").
  The same JSON line carries `weak` (1024 samples per GPU) and `strong_spec_eps8` (BASELINE config 3: 8 speculative line-search step
  sizes per sample) measured in the same run.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

# The hardware threads this process may use, read BEFORE any OpenMP runtime is loaded: with OMP_PROC_BIND set, the runtime pins the
# initial thread to its first place, after which the affinity mask of this thread reads as one CPU.
_AFFINITY0 = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
# (the CPU baseline leg binds the oracle's OpenMP threads; libgomp reads these when it is first loaded -- before torch pulls it in)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")       # one thread per PHYSICAL core first (hardware threads 0 and 1 may be SMT siblings)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak (one datapath, profiles/r01_ubench_fp64_pipe.md)
# every file a counter figure of profiles/traffic.json depends on (tile / general-size / PETS / CE kernels and what they include)
KERNEL_SOURCES = ("kernels.hip", "psweep.h", "sweep_dual.h", "sweep_dual.hip", "device_utils.h", "layout.h", "kernels.h", "wide.hip", "wide.h", "wide16.h", "wide32.h",
                  "ce_device.hip", "ce_device.h", "rat_pow.h", "rat_normal.h")


# ---- the byte / flop model of SURVEY.md section 8(d) (pinned by tests/test_cpu_bench_model.py) --------------------------------
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


def block_bytes(chunk):
    """One rank's contribution to the all-gather (csrc/multi.cpp block_bytes): chunk x (f64 cost + 3 x i32), 8-B aligned."""
    return (chunk * 20 + 7) & ~7


def shard_bounds(B, world, rank):
    """Contiguous theta block [lo, hi) of a rank (ratilqr.jl_amd.distributed.shard_bounds; blocks differ by at most one sample)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def kernel_source_hash():
    """sha256 over the kernel sources: stamps profiles/traffic.json, so that a PMC figure measured on other kernels reads as null."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "ratilqr.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_for(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or (None, why)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, "profiles/traffic.json missing"
    if key not in tj:
        return None, f"no PMC pass recorded for {key}"
    if tj.get("kernels_sha") != kernel_source_hash():
        return None, f"profiles/traffic.json was measured on kernel sources {tj.get('kernels_sha')}, this tree is {kernel_source_hash()}"
    return float(tj[key]), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on kernel sources {tj.get('kernels_sha')} ({tj.get('round', '?')})"


def sq_for(key):
    """SQ-counter digest of a kernel from the committed rocprofv3 --pmc SQ_* passes (profiles/traffic.json, section "sq"; same source
    stamp as the traffic figures), or None: issue_frac = share of the waves' lifetime in which the FP64 datapath issues vector or matrix
    work, parked_frac = SQ_WAIT_ANY share, wait_inst_frac = SQ_WAIT_INST_ANY share."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    if tj.get("kernels_sha") != kernel_source_hash():
        return None
    return tj.get("sq", {}).get(key)


# ---- host cores the CPU baseline may use ---------------------------------------------------------------------------------------
def cgroup_cpu_limit():
    for p in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(p).read().split()[:2]
            if q != "max":
                return float(q) / float(per)
        except Exception:
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except Exception:
        pass
    return None


def host_cpu_info():
    """Hardware threads this process may run on (affinity mask clipped by the cgroup CPU quota) and the physical cores behind them."""
    aff = list(_AFFINITY0)
    lim = cgroup_cpu_limit()
    threads = len(aff) if lim is None else max(1, min(len(aff), int(math.floor(lim + 1e-9))))
    sib = set()
    for c in aff:
        try:
            sib.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip())
        except Exception:
            sib.add(str(c))
    phys = max(1, min(len(sib), threads))
    return {"affinity_threads": len(aff), "cgroup_cpu_limit": lim, "threads": threads, "physical_cores": phys,
            "os_cpu_count": os.cpu_count()}


def cpu_baseline(prob, x0, u0, seconds):
    """The C oracle (reference operation order, -O2, no FMA contraction) on the host cores, one theta-sample per OpenMP thread
    (the reference's one-sample-per-worker layout, cross_entropy...jl:180-191): thread-count sweep, best reported."""
    from oracle import oracle as orc

    P = orc.Problem(prob)
    info = host_cpu_info()
    th1 = draw_theta(24, seed=55)
    t1 = time.perf_counter()
    orc.compute_value_batch(P, x0, u0, th1, nthreads=1)
    v1 = th1.size / (time.perf_counter() - t1)
    cand = sorted({c for c in (info["physical_cores"] // 2, info["physical_cores"], info["threads"]) if c >= 2})
    sweep, per_point = [{"threads": 1, "solves_per_s": v1}], max(1.0, 0.25 * seconds / max(len(cand), 1))

    def run(T, budget, seed0):
        n_done, t_cpu = 0, 0.0
        chunk = max(64, 16 * T)
        while t_cpu < budget:
            th = draw_theta(chunk, seed=seed0 + n_done)
            t0 = time.perf_counter()
            orc.compute_value_batch(P, x0, u0, th, nthreads=T)
            t_cpu += time.perf_counter() - t0
            n_done += chunk
        return n_done, t_cpu

    for T in cand:
        n_done, t_cpu = run(T, per_point, 1000 * T)
        sweep.append({"threads": T, "solves_per_s": n_done / t_cpu})
    best = max(sweep, key=lambda r: r["solves_per_s"])
    n_done, t_cpu = (th1.size, th1.size / v1) if best["threads"] == 1 else run(best["threads"], max(2.0, 0.7 * seconds), 77)
    value = max(best["solves_per_s"], n_done / t_cpu)
    import shutil
    jl = shutil.which("julia")                          # probed at run time: the reference itself is never on the GPU box
    julia_note = ("the Julia reference is not runnable (no julia binary)" if jl is None else
                  f"a julia binary exists ({jl}) but the reference package and its dependencies do not travel to this box")
    return {
        "value": value, "unit": "solves/s", "cores": best["threads"], "kind": "port", "value_1thread": v1,
        "parallel_efficiency": value / (v1 * best["threads"]), "thread_sweep": sweep, "host": info,
        "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
        "sample": f"{n_done} solves of the same workload (theta ~ N(1,2)>0) by the C oracle, OpenMP one sample per thread, best of the "
                  f"thread sweep = {best['threads']} threads ({info['threads']} usable hardware threads on {info['physical_cores']} "
                  f"physical cores: affinity mask clipped by the cgroup quota), {t_cpu:.1f} s; " + julia_note,
    }


# ---- launching ------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (torch.distributed.run, one process per GPU) from this
    process, which never touches the GPU itself (device_count() does not initialise it), and hand their exit code back."""
    import torch

    ndev = torch.cuda.device_count()
    shared = os.environ.get("RATILQR_BENCH_ONE_DEVICE") == "1" or os.environ.get("RATILQR_BENCH_DRY") == "1"    # test hooks
    if ndev < args.gpus and not shared:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} HIP devices, {ndev} visible", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RATILQR_BENCH_SELF_LAUNCHED="1")
    return subprocess.run(cmd, env=env).returncode


class Dist:
    """Process-group plumbing of one rank (RCCL = backend "nccl" on ROCm; gloo and a one-device mapping as test hooks)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("RATILQR_BENCH_BACKEND", "nccl")
        self.dry = os.environ.get("RATILQR_BENCH_DRY") == "1"       # CPU test hook: launch / collective plumbing only, no solver
        if os.environ.get("RATILQR_BENCH_ONE_DEVICE") == "1":
            self.local_rank = 0
        self.multi = self.world > 1 or os.environ.get("RATILQR_BENCH_FORCE_DIST") == "1"
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not self.dry:
            if torch.cuda.device_count() <= self.local_rank:
                raise SystemExit(f"bench.py: rank {self.rank} needs HIP device {self.local_rank}, {torch.cuda.device_count()} visible")
            torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cpu") if self.dry else torch.device("cuda", self.local_rank)
        self.rccl_ranks = 1
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)      # RCCL
            else:
                dist.init_process_group(self.backend)
            # a real collective: every rank contributes its rank id, every rank must see 0 .. world-1
            cdev = self.dev if self.backend == "nccl" else torch.device("cpu")
            ids = torch.empty(self.world, dtype=torch.int64, device=cdev)
            mine = torch.tensor([self.rank], dtype=torch.int64, device=cdev)
            if self.backend == "nccl":
                dist.all_gather_into_tensor(ids, mine)
            else:
                dist.all_gather(list(ids.unbind(0)), mine[0])
            if ids.cpu().tolist() != list(range(self.world)):
                raise SystemExit(f"bench.py: rank census failed: {ids.cpu().tolist()}")
            self.rccl_ranks = dist.get_world_size()

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.multi:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if not self.multi:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, K, W):
        """W untimed steps, then EXACTLY K steps bracketed by barrier + device synchronisation on both sides; max over ranks."""
        for _ in range(W):
            step()
        self.sync()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        self.sync()
        self.barrier()
        self.sync()
        return self.max_over_ranks(time.perf_counter() - t0)


class Workload:
    """One CE batch on this rank: `nloc` theta-samples of a (global) batch `G`, solved by one rat_handle with E speculative step
    sizes, followed -- when there is more than one rank -- by the all-gather of the per-sample costs."""

    def __init__(self, D, prob, x0, u0, theta_local, G, E, kl_bound=0.1, debug=()):
        import ratilqr.jl_amd as rat
        torch = D.torch
        self.D, self.G, self.E, self.kl = D, G, E, kl_bound
        self.nloc = int(theta_local.size)
        self.theta_h = theta_local
        self.ctx = rat.Context(prob, max_batch=max(self.nloc, 1), spec_eps=E, device=D.local_rank)
        for key, val in debug:                            # execution switches (rat_debug_set): the contract leg, A/B profiles
            self.ctx.debug_set(key, val)
        self.ctx.set_initial(x0, u0)
        self.theta = torch.as_tensor(theta_local, dtype=torch.float64, device=D.dev)
        self.chunk = -(-G // D.world)                     # padded shard length of the gather
        # what a rank contributes to the ONE all-gather of a batch (cost + status): a byte block
        # [cost f64 x chunk | status i32 x chunk | iters i32 x chunk | ls_evals i32 x chunk] (rat_multi's layout, csrc/multi.cpp);
        # pad slots of a short shard stay all-ones bytes (NaN / -1) and are never read
        self.bb = block_bytes(self.chunk)
        self.blk = torch.full((self.bb,), 0xFF, dtype=torch.uint8, device=D.dev)
        c = self.chunk
        self.cost = self.blk[: 8 * c].view(torch.float64)
        self.status, self.iters, self.ls = (self.blk[8 * c + 4 * c * q: 8 * c + 4 * c * (q + 1)].view(torch.int32) for q in range(3))
        self.blk_all = torch.empty(D.world * self.bb, dtype=torch.uint8, device=D.dev) if D.multi else self.blk
        # the handle's own HIP stream, seen by torch: the batch, the cost all-gather and the next batch are ordered on it on the
        # device, with no host round trip between steps (the timed region ends with a device synchronisation)
        self.hstream = torch.cuda.ExternalStream(self.ctx.stream, device=D.dev)
        self.fused = None

    def step(self):
        # compute_cost (cross_entropy...jl:173-195): the shard's complete solves and cost = value + kl/theta (:193), all on the
        # device (one kernel launch on the fused path), then the per-sample costs go to every rank
        D, torch = self.D, self.D.torch
        if self.nloc:
            self.ctx.compute_cost_enqueue_ex(self.theta.data_ptr(), self.nloc, self.kl, self.cost.data_ptr(), self.status.data_ptr(),
                                             self.iters.data_ptr(), self.ls.data_ptr())
        if D.multi and D.backend == "nccl":
            with torch.cuda.stream(self.hstream):             # RCCL waits for the batch and the next batch waits for RCCL
                D.dist.all_gather_into_tensor(self.blk_all, self.blk)
        elif D.multi:                                         # host-staged collective (test hook)
            self.hstream.synchronize()
            parts = [torch.empty(self.bb, dtype=torch.uint8) for _ in range(D.world)]
            D.dist.all_gather(parts, self.blk.cpu())
            self.blk_all.copy_(torch.cat(parts))

    def outputs(self):
        """Untimed: per-sample value / status / iteration and line-search counts of the same shard (also checks the timed path's costs)."""
        torch = self.D.torch
        n = max(self.nloc, 1)
        value = torch.empty(n, dtype=torch.float64, device=self.D.dev)
        status, iters, ls = (torch.empty(n, dtype=torch.int32, device=self.D.dev) for _ in range(3))
        if self.nloc:
            self.ctx.solve_batch_dev(self.theta.data_ptr(), self.nloc, value.data_ptr(), status.data_ptr(), iters.data_ptr(), ls.data_ptr())
        torch.cuda.synchronize()
        v = value.cpu().numpy()[: self.nloc]
        assert np.array_equal(self.cost.cpu().numpy()[: self.nloc], v + self.kl / self.theta_h), "compute_cost disagrees with value + kl/theta"
        out = v, status.cpu().numpy()[: self.nloc], iters.cpu().numpy()[: self.nloc], ls.cpu().numpy()[: self.nloc]
        for a, b in zip(out[1:], (self.status, self.iters, self.ls)):
            assert np.array_equal(a, b.cpu().numpy()[: self.nloc]), "the counters written beside the costs differ from rat_ileqg_solve_batch_dev's"
        self.value_t = value
        return out

    def gathered(self):
        """What every rank holds after the gather, as the global batch in order: (cost, status, iters, ls_evals); rank r's block sits at
        [r * block_bytes, ...) and carries its shard length of live entries."""
        D = self.D
        raw = self.blk_all.cpu().numpy()
        c = self.chunk
        cols = ([], [], [], [])
        for r in range(D.world):
            lo, hi = shard_bounds(self.G, D.world, r)
            b = raw[r * self.bb: (r + 1) * self.bb]
            cols[0].append(b[: 8 * c].view(np.float64)[: hi - lo])
            for q in range(3):
                cols[q + 1].append(b[8 * c + 4 * c * q: 8 * c + 4 * c * (q + 1)].view(np.int32)[: hi - lo])
        return tuple(np.concatenate(x) for x in cols)

    def check_gather(self):
        """Every rank holds every shard's costs and counters after the gather."""
        D = self.D
        cost, st, it, ls = self.gathered()
        assert cost.size == self.G and not np.any(np.isnan(cost)), "a block of the gathered costs was never written"
        assert (st >= 0).all() and (it >= 0).all() and (ls >= 0).all(), "a block of the gathered counters was never written"
        lo, hi = shard_bounds(self.G, D.world, D.rank)
        assert np.array_equal(cost[lo:hi], self.cost.cpu().numpy()[: self.nloc]), "all-gather lost this rank's block"
        assert np.array_equal(st[lo:hi], self.status.cpu().numpy()[: self.nloc]), "all-gather lost this rank's statuses"


def roofline_of(w, prof_main, main_kind, iters_h, ls_h):
    """`roofline` of the dominant kernel of workload w from the HIP events recorded inside the timed region.

    `achieved` / `frac` keep SURVEY 8(d)'s contract: ALGORITHMIC bytes of the unfused three-kernel formulation per launch / the launch's
    duration / the 8 TB/s HBM peak.  They are NOT what the kernel moves -- the single-launch solves keep no tile records in HBM -- so the
    object also says, as top-level scalars, what does bound the launch: `bound` = "fp64" (vector + f64 matrix instructions share one
    datapath and issue serially), `issue_frac` (share of a wave's lifetime in which that datapath issues: SQ counters), `fp64_frac`
    (algorithmic flops against the 78.6 TFLOP/s peak) and `hbm_real_gbs` / `hbm_real_frac` (counter traffic / time / peak)."""
    lay = w.ctx.layout_info()
    avg_ms = prof_main["ms"] / max(prof_main["launches"], 1)
    traj_per_launch = prof_main["trajectories"] / max(prof_main["launches"], 1)
    fused = main_kind in ("solve_fused", "solve_block")
    if fused:
        bytes_per_launch = algo_bytes_of_solves(iters_h, ls_h)
        bytes_per_traj = bytes_per_launch / max(len(iters_h), 1)
        kernel_name = ("solve_fused_kernel (one persistent wavefront per theta-sample: whole solve!)" if main_kind == "solve_fused" else
                       "solve_block_kernel (one workgroup per theta-sample -- a wavefront per line-search candidate + a gain-sweep wavefront: whole solve!)")
    else:
        bytes_per_traj = algo_bytes_per_candidate()
        bytes_per_launch = bytes_per_traj * traj_per_launch
        kernel_name = "sweep_kernel<eval> (policy-evaluation Riccati sweep of line-search candidates)"
    sec = avg_ms * 1e-3
    achieved = bytes_per_launch / sec / 1e9 if sec > 0 else 0.0
    key = f"{main_kind}_E{w.E}_B{w.nloc}"
    traffic, tnote = traffic_for(key)
    sq = sq_for(key)
    r = {
        "bound": "fp64" if fused else "hbm",
        "bound_is": ("the FP64 datapath: v_mfma_f64 and f64 vector instructions issue serially on one pipe (profiles/r01_ubench_fp64_pipe.md); "
                     "`achieved` / `frac` are SURVEY 8(d)'s algorithmic-bytes figure (the contract), not bytes moved: see hbm_real_*") if fused
                    else "HBM stream of tile records",
        "kernel": kernel_name,
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "algorithmic_equivalent_gbs": achieved,
        "traffic": traffic, "traffic_source": tnote,
        "hbm_real_gbs": (traffic / sec / 1e9) if (traffic and sec > 0) else None,
        "hbm_real_frac": (traffic / sec / 1e9 / HBM_PEAK_GBS) if (traffic and sec > 0) else None,
        "issue_frac": sq["issue_frac"] if sq else None,
        "parked_frac": sq["parked_frac"] if sq else None,
        "wait_inst_frac": sq["wait_inst_frac"] if sq else None,
        "bytes_per_trajectory": bytes_per_traj, "hbm_record_bytes_per_trajectory": lay["tile_bytes"] + lay["L_bytes"] + 8,
        "trajectories_per_launch": traj_per_launch,
        "avg_launch_ms": avg_ms, "launches": prof_main["launches"],
    }
    if fused:           # the sweep sits at the fp64 ridge -- the FP64 fraction of the same launches
        fl = algo_flops_of_solves(iters_h, ls_h)
        r["fp64_achieved_tflops"] = fl / sec / 1e12
        r["fp64_peak_tflops"] = FP64_PEAK_TFLOPS
        r["fp64_frac"] = fl / sec / 1e12 / FP64_PEAK_TFLOPS
        r["flops_per_solve"] = fl / max(len(iters_h), 1)
    return r


def dry_rank(args, D):
    """CPU test hook (RATILQR_BENCH_DRY=1): the launch, rank census, shard arithmetic, barrier-bracketed timing, max over ranks and
    the cost gather of the N-rank flow with a stand-in for the solver (no GPU in this process; nothing here is a measurement)."""
    torch = D.torch
    G, K, W = args.batch, args.steps, args.warmup
    lo, hi = shard_bounds(G, D.world, D.rank)
    theta = draw_theta(G, seed=1000)
    chunk = -(-G // D.world)
    gathered = {}

    def step():
        cost = torch.full((chunk,), float("nan"), dtype=torch.float64)
        cost[: hi - lo] = torch.as_tensor(0.1 / theta[lo:hi])
        parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(D.world)]
        if D.multi:
            D.dist.all_gather(parts, cost)
        else:
            parts = [cost]
        gathered["all"] = torch.cat([p[: shard_bounds(G, D.world, r)[1] - shard_bounds(G, D.world, r)[0]] for r, p in enumerate(parts)])

    elapsed = D.timed(step, K, W)
    assert np.array_equal(gathered["all"].numpy(), 0.1 / theta), "gathered costs are not the global batch in order"
    if D.rank == 0:
        print(json.dumps({"metric": "iLEQG solves/sec (N=50, n=12, m=4) at CE batch=1024", "value": G * K / elapsed, "unit": "solves/s",
                          "n_gpus": D.world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "DRY RUN (no GPU, no solver): launch and "
                          "collective plumbing only", "dry_run": True, "rccl_ranks": D.rccl_ranks, "collective_backend": D.backend,
                          "self_launched": os.environ.get("RATILQR_BENCH_SELF_LAUNCHED") == "1",
                          "config": {"workload": "dry run", "global_batch": G, "shard_sizes": [shard_bounds(G, D.world, r)[1] - shard_bounds(G, D.world, r)[0] for r in range(D.world)]}}))


def rank_main(args):
    D = Dist(args)
    if D.dry:
        dry_rank(args, D)
        if D.multi:
            D.dist.destroy_process_group()
        return
    torch = D.torch
    import ratilqr.jl_amd as rat

    G, E, K, W = args.batch, args.spec_eps, args.steps, args.warmup
    world, rank = D.world, D.rank
    prob, x0, u0 = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=0)
    theta_global = draw_theta(G, seed=1000)                  # the ONE CE batch every rank knows (same N(0,1) stream on every rank)
    lo, hi = shard_bounds(G, world, rank)
    # The CPU baseline runs FIRST (rank 0, N = 1): the GPU legs then form one contiguous stretch at the end of the run, long enough for
    # an external utilisation sampler to see the device at work.
    cpu = cpu_baseline(prob, x0, u0, args.cpu_seconds) if (rank == 0 and world == 1 and not args.no_cpu) else None

    # ---- primary: the BASELINE configuration -- one CE batch of G = 1024 samples over all ranks (strong scaling) -------------
    dbg = tuple((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.debug)
    w = Workload(D, prob, x0, u0, theta_global[lo:hi], G, E, debug=dbg)
    # conditioning (untimed, before the W warm-up steps): a batch is < 0.5 ms, so W = 3 steps after an idle period are over before the
    # chip has left its idle clocks -- `steady_state` below (>= 1 s of batches) showed the first ~10 ms running 5 % slow.  0.3 s of the
    # same batches first; the timed region is still EXACTLY `steps` batches.
    # (every rank must issue the SAME number of collectives: the number of conditioning blocks is fixed from one block timed on all
    #  ranks and max-reduced, never from a rank-local clock)
    if args.condition_seconds > 0:
        t_blk = D.timed(w.step, 16, 0)
        for _ in range(max(0, min(2000, int(math.ceil(args.condition_seconds / max(t_blk, 1e-6))) - 1))):
            for _ in range(16):
                w.step()
        D.sync()
    for _ in range(W):
        w.step()
    D.sync()
    # untimed pass with HIP events around every kernel kind: per-kernel breakdown (events perturb the stream, so the timed region
    # below only brackets the dominant kernel, whose duration feeds the roofline object)
    w.ctx.profile(True)
    w.ctx.profile_reset()
    n_all = max(2, min(K, 5))
    for _ in range(n_all):
        w.step()
    D.sync()
    prof_all = w.ctx.profile_get()
    main_kind = next((k for k in ("solve_fused", "solve_block") if prof_all[k]["launches"] > 0), "sweep_eval")
    fused = main_kind != "sweep_eval"
    w.ctx.profile(True, kinds=[main_kind])
    w.ctx.profile_reset()
    elapsed = D.timed(w.step, K, 0)
    prof = w.ctx.profile_get()
    w.ctx.profile(False)
    v_h, st_h, it_h, ls_h = w.outputs()
    w.check_gather()
    # statistics of the WHOLE batch, from what the timed step's own all-gather left on this rank (not from this rank's shard)
    _, st_g, it_g, ls_g = w.gathered()
    feasible = float(np.mean((st_g == 0) | (st_g == 3))) if st_g.size else 1.0
    strong = {"value": G * K / elapsed, "unit": "solves/s", "ms_per_step": elapsed / K * 1e3, "global_batch": G,
              "solves_per_gpu": [shard_bounds(G, world, r)[1] - shard_bounds(G, world, r)[0] for r in range(world)], "spec_eps": E}

    # ---- weak scaling and BASELINE config 3 (E = 8) in the same run, N > 1 ---------------------------------------------------------
    weak = strong8 = None
    if world > 1:
        ww = Workload(D, prob, x0, u0, draw_theta(G, seed=1000 + rank), G * world, E)
        ew = D.timed(ww.step, K, W)
        weak = {"value": world * G * K / ew, "unit": "solves/s", "ms_per_step": ew / K * 1e3, "global_batch": world * G,
                "solves_per_gpu": G, "spec_eps": E}
        del ww
        if E != 8 and not args.no_second:
            w8 = Workload(D, prob, x0, u0, theta_global[lo:hi], G, 8)
            K8 = max(3, K // 3)
            e8 = D.timed(w8.step, K8, max(2, W))
            strong8 = {"value": G * K8 / e8, "unit": "solves/s", "ms_per_step": e8 / K8 * 1e3, "steps": K8, "global_batch": G,
                       "spec_eps": 8, "costs_identical_to_primary": bool(torch.equal(w8.cost[: w8.nloc], w.cost[: w.nloc])),
                       # (shards of <= one sample per CU run time-parallel sweeps at E = 1: equal to rounding, not bit for bit)
                       "costs_match_primary_1e-12": bool(torch.allclose(w8.cost[: w8.nloc], w.cost[: w.nloc], rtol=1e-12, atol=0.0, equal_nan=True)),
                       "mean_ls_evals": float(w8.gathered()[3].mean())}
            del w8

    # ---- single-GPU secondaries ---------------------------------------------------------------------------------------------------
    second = nonlin = unfused = steady = large = None
    value = w.value_t
    theta = w.theta
    B = w.nloc
    dev = D.dev
    if world == 1 and not args.no_second:
        # steady state: >= 1 s of back-to-back batches, one HIP event per batch on the handle's stream
        n_ss = int(min(40000, max(200, args.steady_seconds / max(elapsed / K, 1e-5))))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ss + 1)]
        D.sync()
        t0 = time.perf_counter()
        evs[0].record(w.hstream)
        for i in range(n_ss):
            w.step()
            evs[i + 1].record(w.hstream)
        D.sync()
        wall = time.perf_counter() - t0
        per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_ss)])
        steady = {"seconds_requested": args.steady_seconds, "batches": n_ss, "wall_s": wall, "value": B * n_ss / wall, "unit": "solves/s",
                  "per_batch_ms": {"median": float(np.median(per)), "min": float(per.min()), "mean": float(per.mean()),
                                   "p95": float(np.percentile(per, 95)), "max": float(per.max())},
                  "value_from_median": B / (float(np.median(per)) * 1e-3),
                  "note": "per-batch times are HIP-event intervals on the library's stream (launch gaps included); the headline `value` "
                          "is the driver-contract figure over `steps` batches"}
        del evs

    if world == 1 and E != 8 and not args.no_second:
        # E = 8 speculative step sizes per sample (BASELINE config 3's "x 8 line-search eps"); identical results, 8x the candidate
        # work on this problem (every first candidate is accepted).
        ctx8 = rat.Context(prob, max_batch=B, spec_eps=8, device=D.local_rank)
        ctx8.set_initial(x0, u0)
        v8 = torch.empty(B, dtype=torch.float64, device=dev)
        t_c = time.perf_counter()
        while time.perf_counter() - t_c < 0.1:   # conditioning, as for the primary
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        K8 = max(3, K // 3)
        for _ in range(K8):                  # timed WITHOUT profiling events: this path is ~20 launches per batch and an event pair
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())     # around each of them costs it 10 %
        torch.cuda.synchronize()
        e8 = time.perf_counter() - t8
        ctx8.profile(True)                   # per-kernel breakdown from a separate pass
        ctx8.profile_reset()
        for _ in range(K8):
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        torch.cuda.synchronize()
        p8all = ctx8.profile_get()
        k8 = next((k for k in ("solve_fused", "solve_block") if p8all[k]["launches"] > 0), "sweep_eval")
        p8 = p8all[k8]
        a = algo_bytes()
        # spec_eps is an UPPER BOUND on the speculation width (include/ratilqr.h): by default the handle runs the sequential rule, so the
        # candidates EVALUATED -- what the byte model prices (a fraction above 1 means the timed kernels are not doing
        # the priced work) -- are the E = 1 solve's.  Under spec_force every line-search round evaluates 8 candidates per sample
        # (7.08 MB/solve), minus the evaluations the pruning stops -- that leg is reported below without a byte-model fraction.
        width8 = int(ctx8.debug_get("spec_width"))
        rounds = float(np.sum(np.ceil(ls_h / 1.0)))          # E = 1 accepted-candidate count = number of rounds on this workload
        bytes8 = float(B) * a["init"] + float(np.sum(it_h)) * a["sweep_gain"] + float(width8) * rounds * a["candidate"]
        second = {"spec_eps": 8, "spec_width_run": width8,
                  "policy": "spec_eps is an upper bound: the sequential line-search rule on the E = 1 kernels unless spec_force (identical results)",
                  "value": B * K8 / e8, "unit": "solves/s", "ms_per_step": e8 / K8 * 1e3, "steps": K8,
                  "values_identical_to_primary": bool(torch.equal(v8, value)),
                  "algorithmic_GBps": bytes8 * K8 / e8 / 1e9, "algorithmic_bytes_per_solve": bytes8 / B,
                  "kernel_ms_per_step": {k: v["ms"] / K8 for k, v in p8all.items() if v["launches"]},
                  "dominant_kernel": k8}
        # The two streams of a round overlap (candidates 1..7 on one, candidate 0 paired with the next gain sweep on the other), so a
        # per-kernel duration -- and a per-kernel roofline fraction -- is ambiguous; the per-BATCH figure is the robust one: algorithmic bytes
        # of the whole E = 8 batch / its wall time, beside the counter traffic of everything the batch launches (rocprofv3 PMC passes under
        # profiles/).  The candidates carry no tile records on this path, so the bytes moved are far BELOW the algorithmic figure.
        tb8, tn8 = traffic_for(f"batch_E8_B{B}")
        sec8 = e8 / K8
        second["roofline"] = {"bound": "fp64", "scope": "one batch (all kernels, two overlapping streams)",
                              "achieved": bytes8 / sec8 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes8 / sec8 / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_equivalent_gbs": bytes8 / sec8 / 1e9,
                              "traffic": tb8, "traffic_source": tn8,
                              "hbm_real_gbs": tb8 / sec8 / 1e9 if tb8 else None,
                              "hbm_real_frac": tb8 / sec8 / 1e9 / HBM_PEAK_GBS if tb8 else None,
                              "issue_frac_sweep_eval": (sq_for(f"sweep_eval_E8_B{B}") or {}).get("issue_frac")}
        second["hbm_traffic_per_batch"] = tb8
        second["hbm_traffic_per_solve"] = tb8 / B if tb8 else None
        # the same handle forced to its requested width (the speculative kernels: round-based path, pruned): for the record
        ctx8.debug_set("spec_force", 1)
        ctx8.set_initial(x0, u0)
        for _ in range(3):
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        for _ in range(K8):
            ctx8.solve_batch_dev(theta.data_ptr(), B, v8.data_ptr())
        torch.cuda.synchronize()
        e8f = time.perf_counter() - t8
        second["forced_width"] = {"spec_width_run": int(ctx8.debug_get("spec_width")), "path": ctx8.get_path(B), "value": B * K8 / e8f, "unit": "solves/s",
                                  "ms_per_step": e8f / K8 * 1e3, "values_identical_to_primary": bool(torch.equal(v8, value))}
        del ctx8

    if world == 1 and not args.no_second:
        # a nonlinear workload (cubic drift kappa x^3 on the same tables): more iterations per solve and theta-dependent
        # iteration counts, i.e. divergent per-sample control flow inside one launch
        probn, x0n, u0n = rat.synthetic_lq_problem(n=12, m=4, N=50, seed=0, kappa=0.05)
        ctxn = rat.Context(probn, max_batch=B, spec_eps=E, device=D.local_rank)
        ctxn.set_initial(x0n, u0n)
        vn = torch.empty(B, dtype=torch.float64, device=dev)
        stn, itn, lsn = (torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
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

        # CE batches larger than the chip's 1024 SIMDs: two samples per SIMD (the 256-register, tile-free variant of the fused kernel: the default
        # there since round 5, switch fused_occ2 = -1)
        large = {}
        for Bl in (2048, 4096, 8192):
            ctxl = rat.Context(prob, max_batch=Bl, spec_eps=E, device=D.local_rank)
            ctxl.set_initial(x0, u0)
            thl = torch.as_tensor(draw_theta(Bl, seed=4242), dtype=torch.float64, device=dev)
            cl = torch.empty(Bl, dtype=torch.float64, device=dev)
            t_c = time.perf_counter()                    # (conditioning as for the primary: the first launches of a fresh handle touch its
            while time.perf_counter() - t_c < 0.1:       #  tile pool for the first time)
                ctxl.compute_cost_dev(thl.data_ptr(), Bl, 0.1, cl.data_ptr())
            torch.cuda.synchronize()
            tl = time.perf_counter()
            Kl = max(5, K // 2)
            for _ in range(Kl):
                ctxl.compute_cost_enqueue(thl.data_ptr(), Bl, 0.1, cl.data_ptr())
            torch.cuda.synchronize()
            el = time.perf_counter() - tl
            large[str(Bl)] = {"value": Bl * Kl / el, "unit": "solves/s", "ms_per_step": el / Kl * 1e3, "steps": Kl}
            del ctxl

    contract = ce_sec = host_sec = e8_shard = None
    if world == 1 and fused and E == 1 and not args.no_second:
        # SURVEY 8(d) to the letter: "tiles must be materialised per trajectory per step" and every sample runs its own initialize! --
        # the same batch on the same kernel with its tile records put back (rollouts write 3.4 KB per step, sweeps load them) and the
        # shared initial trajectory switched off.  Bit-identical costs; the number the headline's tile-free / shared-init form is an
        # optimisation OF.
        wc = Workload(D, prob, x0, u0, theta_global[lo:hi], G, E, debug=(("materialize", 1), ("init_share", 0)))
        ec = D.timed(wc.step, K, max(W, 3))
        trc, _ = traffic_for(f"solve_fused_mat_E{E}_B{wc.nloc}")
        contract = {"what": "same batch, same kernel, switches materialize = 1 and init_share = 0: tile records in HBM (written by the rollouts, "
                            "loaded by the sweeps), initialize! rolled out by every sample", "value": G * K / ec, "unit": "solves/s",
                    "ms_per_step": ec / K * 1e3, "steps": K, "costs_identical_to_primary": bool(torch.equal(wc.cost[: wc.nloc], w.cost[: w.nloc])),
                    "algorithmic_GBps": algo_bytes_of_solves(it_h, ls_h) * K / ec / 1e9,
                    "frac_of_hbm_peak_on_algorithmic_bytes": algo_bytes_of_solves(it_h, ls_h) * K / ec / 1e9 / HBM_PEAK_GBS,
                    "traffic": trc, "hbm_real_frac": (trc / (ec / K) / 1e9 / HBM_PEAK_GBS) if trc else None}
        del wc

    if world == 1 and not args.no_second:
        # The user-facing calls, host buffers in and out (what the reference's signatures are):
        #  * compute_cost (cross_entropy...jl:173-195): theta[1024] on the host -> cost[1024] on the host;
        #  * one RAT iLQR solve! (:364-415): 5 CE iterations of 1024 samples (draws, elite selection, mu / sigma update) + the final
        #    iLEQG solve at theta_opt returning x / l / L -- what a receding-horizon controller pays per control step.
        from ratilqr.jl_amd import cross_entropy as cemod
        ces = rat.CrossEntropyBilevelOptimizationSolver(num_samples=B, num_elite=max(3, B // 10), spec_eps=E, device=D.local_rank)
        th_host = np.array(theta_global[:B])
        for _ in range(5):
            c_host = cemod.compute_cost(ces, prob, x0, u0, th_host, 0.1)
        reps = 100
        th0 = time.perf_counter()
        for _ in range(reps):
            c_host = cemod.compute_cost(ces, prob, x0, u0, th_host, 0.1)
        eh = (time.perf_counter() - th0) / reps
        host_sec = {"what": "compute_cost with host arrays in and out (one H2D of theta, one D2H of the costs, one host wait per call)",
                    "ms_per_call": eh * 1e3, "value": B / eh, "unit": "solves/s", "calls": reps,
                    "costs_identical_to_primary": bool(np.array_equal(c_host, w.cost[:B].cpu().numpy())),
                    "device_resident_ms_per_step": strong["ms_per_step"]}
        for rep in range(3):
            cemod.solve_(ces, prob, x0, u0, 1234 + rep, kl_bound=0.1)
        ts = []
        for rep in range(50):
            t0 = time.perf_counter()
            out_ce = cemod.solve_(ces, prob, x0, u0, 99 + rep, kl_bound=0.1)
            ts.append(time.perf_counter() - t0)
        cctx = ces.context(prob)
        cctx.profile(True)
        cctx.profile_reset()
        cemod.solve_(ces, prob, x0, u0, 7, kl_bound=0.1)
        prc = {k: v for k, v in cctx.profile_get().items() if v["launches"]}
        cctx.profile(False)
        kms = sum(v["ms"] for v in prc.values())
        n_solves_ce = int(ces.c.n_solves)
        ts = np.array(ts)
        ce_sec = {"what": "one CrossEntropyBilevelOptimizationSolver solve! (rat_ce_solve): 5 CE iterations x 1024 samples + the final solve at "
                          "theta_opt; host x_0 / u_array in, (theta_opt, x, l, L, value) out; built-in generator",
                  "ms_per_solve": {"median": float(np.median(ts)) * 1e3, "min": float(ts.min()) * 1e3, "p95": float(np.percentile(ts, 95)) * 1e3},
                  "calls": int(ts.size), "kernel_ms_sum": kms, "kernel_launches": int(sum(v["launches"] for v in prc.values())),
                  "host_share": 1.0 - kms / (float(np.median(ts)) * 1e3) if kms > 0 else None,
                  "ileqg_solves_per_call": 5 * B + 1,
                  "solves_per_s_equivalent": (5 * B + 1) / float(np.median(ts)), "theta_opt": float(out_ce[0])}
        del ces, n_solves_ce

    if world == 1 and E != 8 and not args.no_second:
        # BASELINE config 3's shard on 8 GPUs: 128 samples x 8 speculative step sizes in one launch
        c8s = rat.Context(prob, max_batch=128, spec_eps=8, device=D.local_rank)
        c8s.set_initial(x0, u0)
        th8 = torch.as_tensor(theta_global[:128], dtype=torch.float64, device=dev)
        co8 = torch.empty(128, dtype=torch.float64, device=dev)
        t_c = time.perf_counter()
        while time.perf_counter() - t_c < 0.1:
            c8s.compute_cost_dev(th8.data_ptr(), 128, 0.1, co8.data_ptr())
        K8s = max(20, K)
        torch.cuda.synchronize()
        t8s = time.perf_counter()
        for _ in range(K8s):
            c8s.compute_cost_enqueue(th8.data_ptr(), 128, 0.1, co8.data_ptr())
        torch.cuda.synchronize()
        e8s = (time.perf_counter() - t8s) / K8s
        e8_shard = {"what": "config 3's per-GPU shard at 8 GPUs: 128 samples, handle created with 8 speculative step sizes (an upper bound: the "
                            "sequential rule runs unless spec_force)", "spec_width_run": int(c8s.debug_get("spec_width")), "ms_per_batch": e8s * 1e3, "steps": K8s,
                    "path": c8s.get_path(128), "costs_identical_to_primary": bool(torch.equal(co8, w.cost[:128])),
                    "costs_match_primary_1e-12": bool(torch.allclose(co8, w.cost[:128], rtol=1e-12, atol=0.0))}
        c8s.debug_set("spec_force", 1)                        # the speculative workgroup-per-sample kernel (E = 8 in one launch), for the record
        c8s.set_initial(x0, u0)
        for _ in range(3):
            c8s.compute_cost_dev(th8.data_ptr(), 128, 0.1, co8.data_ptr())
        torch.cuda.synchronize()
        t8s = time.perf_counter()
        for _ in range(K8s):
            c8s.compute_cost_enqueue(th8.data_ptr(), 128, 0.1, co8.data_ptr())
        torch.cuda.synchronize()
        e8_shard["forced_width_ms_per_batch"] = (time.perf_counter() - t8s) / K8s * 1e3
        del c8s

    shard_lat = pets_sec = nm_sec = wide_sec = None
    if world == 1 and not args.no_second:
        # What each rank of an N-GPU run of the BASELINE metric executes: ONE launch over 1024 / N samples.  A strong-scaling shard costs
        # one solve's latency whatever its size, so these driver-timed figures are the scaling curve's proxy when no multi-GPU node is
        # available: value(N) ~ 1024 / (shard_latency_ms(1024 / N) + all-gather).
        shard_lat = {}
        for Bs in (512, 256, 128):
            cs = rat.Context(prob, max_batch=Bs, spec_eps=E, device=D.local_rank)
            cs.set_initial(x0, u0)
            ths = torch.as_tensor(theta_global[:Bs], dtype=torch.float64, device=dev)
            cst = torch.empty(Bs, dtype=torch.float64, device=dev)
            t_c = time.perf_counter()
            while time.perf_counter() - t_c < 0.1:
                cs.compute_cost_dev(ths.data_ptr(), Bs, 0.1, cst.data_ptr())
            Ks = max(20, K)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(Ks):
                cs.compute_cost_enqueue(ths.data_ptr(), Bs, 0.1, cst.data_ptr())
            torch.cuda.synchronize()
            es = time.perf_counter() - ts
            shard_lat[str(Bs)] = {"ms_per_batch": es / Ks * 1e3, "steps": Ks, "path": cs.get_path(Bs),
                                  "time_parallel_sweeps": bool(cs.debug_get("block_psw")) and Bs <= 256,
                                  "costs_identical_to_primary": bool(torch.equal(cst, w.cost[:Bs])),
                                  # (shards of <= one sample per CU run time-parallel sweeps: equal to rounding, not bit for bit)
                                  "costs_match_primary_1e-12": bool(torch.allclose(cst, w.cost[:Bs], rtol=1e-12, atol=0.0, equal_nan=True))}
            del cs

        # BASELINE config 5: PETS forward simulation (src/pets.jl:128-157), 100 control samples x 100 stochastic rollouts = 10k
        # trajectories, N = 30, n = 12, m = 4, cubic drift, Gaussian process noise from the device Philox generator
        from ratilqr.jl_amd import pets
        rp = np.random.default_rng(8)
        Ap = 0.9 * np.linalg.qr(rp.standard_normal((12, 12)))[0]
        Bp = rp.standard_normal((12, 4)) / np.sqrt(12)
        gprob = rat.LQGenerativeProblem(Ap, Bp, 30, ("gaussian", np.zeros(12), 0.03 * np.eye(12)), Q=np.eye(12), R=0.1 * np.eye(4), Qf=np.eye(12),
                                        kappa=-0.01)
        xp0 = rp.standard_normal(12)
        pets_sec = {"workload": "PETS compute_cost (pets.jl:128-157): stochastic rollouts with running cost, N=30, n=12, m=4, cubic drift, Gaussian "
                                "noise from the device Philox generator; host buffers in and out", "runs": {}}
        for S_, K_ in ((100, 100), (1000, 1000)):
            ds = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([np.eye(4)] * 30), num_control_samples=S_,
                                                          num_trajectory_samples=K_, device=D.local_rank)
            ctrl = 0.3 * rp.standard_normal((S_, 30, 4))
            for _ in range(3):
                cpets = pets.compute_cost_serial(ds, gprob, xp0, ctrl, None, False, seed=11)
            pctx = ds.context(gprob)
            reps = 50 if S_ * K_ <= 100000 else 20
            tp = time.perf_counter()
            for i in range(reps):
                cpets = pets.compute_cost_serial(ds, gprob, xp0, ctrl, None, False, seed=11 + i)
            ep = (time.perf_counter() - tp) / reps
            pctx.profile(True, kinds=["pets"])               # kernel time by events in a second pass: the records cost ~8 us per call
            pctx.profile_reset()
            for i in range(reps):
                pets.compute_cost_serial(ds, gprob, xp0, ctrl, None, False, seed=11 + i)
            pk = pctx.profile_get()["pets"]
            pctx.profile(False)
            assert np.all(np.isfinite(cpets))
            kern_ms = pk["ms"] / max(pk["launches"], 1)
            # per trajectory step: stage cost 2 x 16 x 16 + dynamics 2 x 12 x 16 + noise factor 2 x 12 x 12 flops (the generator's integer
            # rounds and the Box-Muller transcendentals are not counted)
            fl = S_ * K_ * 30 * (2 * 16 * 16 + 2 * 12 * 16 + 2 * 12 * 12)
            pets_sec["runs"][f"{S_}x{K_}"] = {
                "control_samples": S_, "rollouts_per_sample": K_, "trajectories": S_ * K_, "ms_per_call": ep * 1e3,
                "trajectories_per_s": S_ * K_ / ep, "steps_per_s": S_ * K_ * 30 / ep, "kernel_ms": kern_ms,
                "kernel_trajectories_per_s": S_ * K_ / (kern_ms * 1e-3) if kern_ms > 0 else None,
                "roofline": {"bound": "fp64 datapath (per step: 11-14 f64 MFMAs for 16 trajectories as MFMA columns + Philox and Box-Muller "
                                      "on the vector ALU; no HBM stream: 1.3 KB in, 8 B out per trajectory)",
                             "achieved": fl / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": fl / (kern_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if kern_ms > 0 else None}}
            del ds
        pets_sec["value"] = pets_sec["runs"]["100x100"]["trajectories_per_s"]
        pets_sec["unit"] = "trajectories/s (BASELINE config 5: 10k trajectories per call)"
        # one PETS solve! (pets.jl:270-281): 5 CE iterations of 100 control samples x 100 rollouts, the loop over control sequences resident on
        # the device (sampling, rollouts, elites and the smoothed update in one enqueue chain, one host wait); the library call alone
        # (control normals drawn on the device: zc = NULL) and the kernel sum from HIP events
        import ctypes as _C
        from ratilqr.jl_amd import _native as _nv
        dsv = rat.CrossEntropyDirectOptimizationSolver(np.zeros((30, 4)), np.stack([0.3 * np.eye(4)] * 30), num_control_samples=100,
                                                       num_trajectory_samples=100, num_elite=10, iter_max=5, device=D.local_rank)
        sctx = dsv.context(gprob)
        xpp = _nv.f64(xp0)

        def one_solve(seed):
            _nv.check(_nv.lib().rat_pets_solve(sctx.h, _C.byref(dsv.c), _nv.P(xpp), 0, None, None, None, _C.c_uint64(seed)))

        for i in range(5):
            one_solve(20 + i)
        tsv = []
        for i in range(30):
            t0 = time.perf_counter(); one_solve(40 + i); tsv.append(time.perf_counter() - t0)
        sctx.profile(True); sctx.profile_reset()
        one_solve(99)
        pv = sctx.profile_get(); sctx.profile(False)
        pets_sec["solve"] = {"what": "rat_pets_solve, 5 iterations x (100 control samples x 100 rollouts), device-resident loop, control normals and "
                                     "rollout noise from the device generators", "ms_per_solve": float(np.median(tsv)) * 1e3,
                             "kernel_sum_ms": float(sum(v["ms"] for v in pv.values())), "launches": int(sum(v["launches"] for v in pv.values())),
                             "mu_finite": bool(np.all(np.isfinite(dsv.mu_array)))}
        del dsv

        # BASELINE config 4: RAT iLQR++ (src/nelder_mead_bilevel_optimization.jl:276-352) on the headline problem: every Nelder-Mead
        # iteration's vertices -- and those of the iterations after (three deep in the first call, two afterwards) -- are evaluated ahead in
        # one batched device call (driver.cpp: nm_tree / nm_plan)
        from ratilqr.jl_amd import nelder_mead as nm
        nms = rat.NelderMeadBilevelOptimizationSolver(device=D.local_rank)

        def nm_fresh_solve():
            # a FRESH solver's solve! each time (the reference keeps c_high / c_low and the shrunk theta_*_init across solve! calls,
            # nelder_mead...jl:164-168,283,294 -- reproduced by the library, so the state is put back by hand; the device context is reused)
            nms.c.has_c_high = nms.c.has_c_low = 0
            nms.c.theta_high_init, nms.c.theta_low_init = 3.0, 1e-8
            ns0, nb0 = int(nms.c.n_solves), int(nms.c.n_batches)
            r = nm.solve_(nms, prob, x0, u0, 0.1)
            return r, int(nms.c.n_solves) - ns0, int(nms.c.n_batches) - nb0

        for _ in range(2):
            nm_fresh_solve()
        reps = 5
        tn0 = time.perf_counter()
        for _ in range(reps):
            (th_nm, _, _, _, val_nm), n_seq, n_bat = nm_fresh_solve()
        enm = (time.perf_counter() - tn0) / reps
        nm_sec = {"workload": "one NelderMeadBilevelOptimizationSolver solve! of a fresh solver (defaults) on the headline problem, kl_bound = 0.1",
                  "ms_per_solve": enm * 1e3, "nm_iterations": int(nms.c.iter_current), "batched_device_calls": n_bat,
                  "sequential_ileqg_solves_replaced": n_seq, "theta_opt": th_nm, "objective": val_nm,
                  "ms_per_sequential_solve": enm * 1e3 / max(n_seq, 1)}

        # general sizes (wide.hip; ileqg.jl:229 takes the dimensions from the arrays): CE batch of 1024 at 16 x 4 (wide16.h), 24 x 8 and 32 x 32 (wide32.h)
        wide_sec = {"workload": "rat_ileqg_solve_batch, LQ-plus-noise problems of the headline recipe beyond the 12 + 4 tile, N = 50, CE batch 1024, "
                                "host arrays in and out", "runs": {}}
        for n_, m_ in ((16, 4), (24, 8), (32, 32)):
            wprob, wx0, wu = rat.synthetic_lq_problem(n=n_, m=m_, N=50)
            wth = np.abs(1.0 + 2.0 * np.random.default_rng(1).standard_normal(1024)) * 0.2
            wctx = rat.Context(wprob, max_batch=1024, device=D.local_rank)
            wv, wst, wit, _ = wctx.solve_batch(wx0, wu, wth)
            tw = []
            for _ in range(3):
                t0w = time.perf_counter()
                wctx.solve_batch(wx0, wu, wth)
                tw.append(time.perf_counter() - t0w)
            wide_sec["runs"][f"{n_}x{m_}"] = {"ms_per_batch": min(tw) * 1e3, "solves_per_s": 1024 / min(tw), "feasible": int((wst == 0).sum()),
                                              "iterations_max": int(wit.max())}
            del wctx
        wide_sec["value"] = wide_sec["runs"]["16x4"]["solves_per_s"]
        wide_sec["unit"] = "solves/s (n = 16, m = 4)"

    if fused and world == 1 and not args.no_second:
        # per-phase breakdown of the same batch on the round-based path (one launch per phase; what the fused kernel replaces)
        ctxu = rat.Context(prob, max_batch=B, spec_eps=E, device=D.local_rank)
        ctxu.set_path("rounds")
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
        out = {
            "metric": "iLEQG solves/sec (N=50, n=12, m=4) at CE batch=1024",
            "value": strong["value"],
            "unit": "solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": strong["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong",
            "value_is": "strong: ONE CE batch of 1024 theta-samples (BASELINE.json metric) sharded over the ranks in contiguous blocks, "
                        "cost all-gather included; `weak` = 1024 samples per GPU",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rccl_ranks": D.rccl_ranks,
            "collective_backend": ("rccl (torch.distributed nccl)" if D.backend == "nccl" else D.backend) if D.multi else None,
            "self_launched": os.environ.get("RATILQR_BENCH_SELF_LAUNCHED") == "1",
            "config": {
                "workload": "batched iLEQG solves of one CE batch (compute_cost): synthetic LQ-plus-noise, N=50, n=12, m=4, "
                            "W=1e-3 I, theta ~ N(1,2)>0, kl=0.1, iLEQG defaults",
                "global_batch": G, "ce_batch_per_gpu": strong["solves_per_gpu"], "spec_eps": E,
                "parallelism": f"theta-shards x{world} (contiguous blocks), one cost all-gather per batch" if world > 1 else "single GPU",
                "feasible_fraction": feasible, "mean_iters": float(it_g.mean()), "mean_ls_evals": float(ls_g.mean()),
                "statistics_from": "the status / iteration / line-search counts gathered with the costs by the timed step's one collective",
                # what the timed kernel does NOT do that SURVEY 8(d)'s byte model prices (both bit-identical to doing it; the
                # `secondary_contract` leg measures the same batch with both switched back):
                "tile_free": bool(fused and not w.ctx.debug_get("materialize")),
                "init_shared": bool(fused and w.ctx.debug_get("init_share")),
                "tile_free_means": "no tile records in HBM: rollouts store [x_t; u_t] and the [c_x | c_u | c] row, every sweep forms f_x | f_u and "
                                   "the cost Hessian of step t in registers from x_t and the problem tables (per trajectory, per step)",
                "init_shared_means": "initialize!'s open-loop rollout depends on (x_0, u_array) only, not on theta: rolled out once per "
                                     "rat_set_initial, copied by the samples",
                "path": w.ctx.get_path(max(w.nloc, 1)),
            },
            "strong": strong,
            "roofline": roofline_of(w, prof[main_kind], main_kind, it_h, ls_h),
            "kernel_ms_per_step": {k: v["ms"] / n_all for k, v in prof_all.items()},
        }
        # one scalar per secondary at the top level (a parser that keeps only scalars still sees every leg)
        flat = {
            "contract_solves_per_s": contract["value"] if contract else None,
            "e8_solves_per_s": second["value"] if second else None,
            "e8_forced_solves_per_s": second["forced_width"]["value"] if second else None,
            "e8_shard128_ms": e8_shard["ms_per_batch"] if e8_shard else None,
            "e8_forced_shard128_ms": e8_shard["forced_width_ms_per_batch"] if e8_shard else None,
            "nonlinear_solves_per_s": nonlin["value"] if nonlin else None,
            "shard512_ms": shard_lat["512"]["ms_per_batch"] if shard_lat else None,
            "shard256_ms": shard_lat["256"]["ms_per_batch"] if shard_lat else None,
            "shard128_ms": shard_lat["128"]["ms_per_batch"] if shard_lat else None,
            "ce_solve_ms": ce_sec["ms_per_solve"]["median"] if ce_sec else None,
            "compute_cost_host_ms": host_sec["ms_per_call"] if host_sec else None,
            "nm_ms_per_solve": nm_sec["ms_per_solve"] if nm_sec else None,
            "wide_16x4_solves_per_s": wide_sec["runs"]["16x4"]["solves_per_s"] if wide_sec else None,
            "wide_24x8_solves_per_s": wide_sec["runs"]["24x8"]["solves_per_s"] if wide_sec else None,
            "wide_32x32_solves_per_s": wide_sec["runs"]["32x32"]["solves_per_s"] if wide_sec else None,
            "pets_traj_per_s": pets_sec["value"] if pets_sec else None,
            "pets_1m_traj_per_s": pets_sec["runs"]["1000x1000"]["trajectories_per_s"] if pets_sec else None,
            "pets_solve_ms": pets_sec["solve"]["ms_per_solve"] if pets_sec else None,
            "steady_solves_per_s": steady["value"] if steady else None,
            "batch2048_solves_per_s": large["2048"]["value"] if large else None,
            "batch4096_solves_per_s": large["4096"]["value"] if large else None,
            "batch8192_solves_per_s": large["8192"]["value"] if large else None,
        }
        out.update({k: v for k, v in flat.items() if v is not None})
        # ... and inside `roofline`, which the driver's record keeps whole: the SURVEY 8(d)-to-the-letter contract figure
        # with its fraction, the strong-scaling shard latencies and the user-facing calls; `frac_of_bound` = the fraction of the roofline
        # `bound` names (fp64), beside `frac` = SURVEY 8(d)'s algorithmic-bytes figure against the HBM peak
        rf = out["roofline"]
        rf["frac_of_bound"] = rf.get("fp64_frac") if rf.get("bound") == "fp64" else rf.get("frac")
        rf["frac_is"] = "SURVEY 8(d) contract: algorithmic bytes / launch time / HBM peak (NOT bytes moved); the binding roofline's fraction is frac_of_bound"
        if contract:
            rf["contract_frac"] = contract["frac_of_hbm_peak_on_algorithmic_bytes"]
            rf["contract_hbm_real_frac"] = contract["hbm_real_frac"]
        rf.update({k: v for k, v in flat.items() if v is not None})
        # key ORDER: the driver's record keeps the first 24 keys of `roofline` -- the contract's six, then the scalars a
        # reader of BENCH_rNN.json needs; the prose (`bound_is`, `traffic_source`, `frac_is`, `kernel`) goes last
        front = ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_of_bound", "contract_solves_per_s", "contract_frac",
                 "shard128_ms", "shard256_ms", "shard512_ms", "ce_solve_ms", "e8_solves_per_s", "e8_shard128_ms", "nm_ms_per_solve",
                 "fp64_frac", "issue_frac", "hbm_real_frac", "avg_launch_ms", "batch4096_solves_per_s", "wide_32x32_solves_per_s",
                 "pets_solve_ms", "contract_hbm_real_frac")
        prose = ("kernel", "bound_is", "frac_is", "traffic_source")
        out["roofline"] = rf = {**{k: rf[k] for k in front if k in rf}, **{k: v for k, v in rf.items() if k not in front and k not in prose},
                                **{k: rf[k] for k in prose if k in rf}}
        if weak is not None:
            out["weak"] = weak
        if strong8 is not None:
            out["strong_spec_eps8"] = strong8
        if steady is not None:
            out["steady_state"] = steady
        if unfused is not None:
            out["round_based_path"] = unfused
        if second is not None:
            out["secondary_spec_eps8"] = second
        if nonlin is not None:
            out["secondary_nonlinear"] = nonlin
        if large:
            out["secondary_large_batch"] = large
        for key, val in (("secondary_contract", contract), ("secondary_compute_cost_host", host_sec), ("secondary_ce_solve", ce_sec),
                         ("secondary_spec_eps8_shard128", e8_shard),
                         ("shard_latency_ms", shard_lat), ("secondary_pets", pets_sec), ("secondary_nm", nm_sec), ("secondary_wide", wide_sec)):
            if val is not None:
                out[key] = val
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if D.multi:
        D.dist.destroy_process_group()


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="the CE batch (global: sharded over the ranks)")
    ap.add_argument("--spec-eps", type=int, default=1, help="E speculative line-search step sizes per sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-second", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="execution switch of the primary workload's handle (rat_debug_set), e.g. --debug materialize=1 --debug init_share=0")
    ap.add_argument("--steady-seconds", type=float, default=4.0, help="length of the steady_state leg (back-to-back batches)")
    ap.add_argument("--condition-seconds", type=float, default=0.3, help="untimed batches before the warm-up steps (clock ramp)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))             # (nothing has touched the GPU in this process)
    rank_main(args)


if __name__ == "__main__":
    main()
