"""bench.py as a launcher (no GPU needed): `python bench.py --gpus N` WITHOUT a torchrun wrapper starts the N ranks itself, refuses
to run on fewer devices than ranks, and never reports more ranks than took part in a real collective.

The RATILQR_BENCH_DRY hook replaces the solver by a stand-in (there is no GPU here and the product has no CPU path); everything around
it -- self-launch through torch.distributed.run, rank census by all-gather, contiguous theta shards of the ONE global CE batch,
barrier-bracketed timing, max over ranks, the cost gather, rank 0's JSON line -- is bench.py's real code over gloo."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
spec = importlib.util.spec_from_file_location("bench", BENCH)
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _run(args, env_extra, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("RATILQR_BENCH")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=600)


def _json(out):
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_gpus_2_self_launches_two_ranks_without_torchrun():
    d = _json(_run(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"RATILQR_BENCH_DRY": "1", "RATILQR_BENCH_BACKEND": "gloo"}))
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["self_launched"] is True and d["dry_run"] is True
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 1024 and d["config"]["shard_sizes"] == [512, 512]
    assert d["steps"] == 4 and d["value"] > 0 and abs(d["value"] - 1024 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]


def test_the_drivers_own_eight_rank_command_line():
    """exactly how the round-end scaling bench is launched: torch.distributed.run with one rank per GPU (here: gloo + the stand-in solver)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") and not k.startswith("RATILQR_BENCH")}
    env.update({"RATILQR_BENCH_DRY": "1", "RATILQR_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True,
                         timeout=900)
    d = _json(out)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["self_launched"] is False
    assert d["scaling"] == "strong" and d["config"]["shard_sizes"] == [128] * 8 and d["steps"] == 3
    assert len([ln for ln in out.stdout.splitlines() if ln.startswith("{")]) == 1          # rank 0 prints ONE line


def test_ragged_global_batch_over_three_ranks():
    d = _json(_run(["--gpus", "3", "--steps", "2", "--warmup", "0", "--batch", "1000"], {"RATILQR_BENCH_DRY": "1", "RATILQR_BENCH_BACKEND": "gloo"}))
    assert d["rccl_ranks"] == 3 and d["config"]["shard_sizes"] == [334, 333, 333]


def test_more_ranks_than_devices_fails_loudly():
    """No hook: this box has no GPU, so --gpus 2 must exit non-zero before anything is launched (round 1 silently ran one rank)."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert out.returncode != 0 and "HIP devices" in out.stderr and "{" not in out.stdout


def test_world_size_that_contradicts_gpus_fails():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "RATILQR_BENCH_DRY": "1"}, drop=())
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_shard_bounds_match_the_library():
    from ratilqr.jl_amd import distributed as rd
    for B, w in ((1024, 8), (1000, 3), (5, 8), (1024, 1)):
        blocks = [bench.shard_bounds(B, w, r) for r in range(w)]
        assert blocks == [rd.shard_bounds(B, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == B and all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))


def test_host_cpu_info_respects_affinity_and_quota(monkeypatch):
    info = bench.host_cpu_info()
    assert 1 <= info["threads"] <= info["affinity_threads"] <= (os.cpu_count() or 1) and 1 <= info["physical_cores"] <= info["threads"]
    monkeypatch.setattr(bench, "cgroup_cpu_limit", lambda: 2.5)
    assert bench.host_cpu_info()["threads"] == min(2, info["affinity_threads"])
    monkeypatch.setattr(bench, "cgroup_cpu_limit", lambda: None)
    monkeypatch.setattr(bench, "_AFFINITY0", [bench._AFFINITY0[0]])
    assert bench.host_cpu_info()["threads"] == 1


def test_traffic_is_null_when_the_kernels_changed(tmp_path, monkeypatch):
    """roofline.traffic comes from committed PMC passes: it must read null once the kernel sources differ from the measured ones."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    src = tmp_path / "ratilqr.jl_amd" / "csrc"
    src.mkdir(parents=True)
    for f in bench.KERNEL_SOURCES:
        (src / f).write_text("// " + f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    sha = bench.kernel_source_hash()
    (prof / "traffic.json").write_text(json.dumps({"kernels_sha": sha, "round": "r02", "solve_fused_E1_B1024": 1.2e9}))
    assert bench.traffic_for("solve_fused_E1_B1024")[0] == 1.2e9
    assert bench.traffic_for("solve_fused_E1_B512")[0] is None
    (src / "kernels.hip").write_text("// edited")
    val, why = bench.traffic_for("solve_fused_E1_B1024")
    assert val is None and sha in why
