"""Prints the headline and the secondaries of a bench.py JSON line.   python tools/bench_brief.py file.json
   python tools/bench_brief.py --update-design [profiles/r06_bench_default.json [profiles/r05_bench_default.json]]
rewrites the block between <!-- BENCH:BEGIN --> and <!-- BENCH:END --> of DESIGN.md (section 5's numbers) from the committed bench line of the
round, with the previous round's beside it and the rocprofv3 kernel averages of profiles/<round>_kernel_stats_*.csv: no figure of that block
is copied by hand."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def brief(path):
    d = load(path)
    print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 4), "path", d["config"].get("path"))
    for k in ("contract_solves_per_s", "e8_solves_per_s", "e8_forced_solves_per_s", "e8_shard128_ms", "e8_forced_shard128_ms", "nonlinear_solves_per_s", "shard512_ms",
              "shard256_ms", "shard128_ms", "ce_solve_ms", "compute_cost_host_ms", "nm_ms_per_solve", "wide_16x4_solves_per_s", "wide_24x8_solves_per_s",
              "wide_32x32_solves_per_s", "pets_traj_per_s", "pets_1m_traj_per_s", "pets_solve_ms", "steady_solves_per_s"):
        print("  ", k, d.get(k))
    r = d["roofline"]
    print("roofline:", {k: r.get(k) for k in ("bound", "frac", "frac_of_bound", "fp64_frac", "issue_frac", "hbm_real_frac", "contract_frac", "traffic", "avg_launch_ms")})
    print("cpu_baseline:", d.get("cpu_baseline"))


def kernel_avg_us(rnd, cfg, pat):
    f = os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_{cfg}.csv")
    if not os.path.exists(f):
        return None
    best = None
    for row in csv.DictReader(open(f)):
        if pat in row["Name"] and (best is None or int(row["Calls"]) > best[1]):
            best = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    return best[0] if best else None


def fmt(v, kind):
    if v is None:
        return "–"
    if kind == "M":
        return f"{v / 1e6:.3f} M"
    if kind == "k":
        return f"{v / 1e3:.1f} k"
    if kind == "ms":
        return f"{v:.4f} ms"
    if kind == "frac":
        return f"{v:.3f}"
    return str(v)


def design_block(cur_path, prev_path):
    d, p = load(cur_path), (load(prev_path) if prev_path and os.path.exists(prev_path) else {})
    rnd = re.search(r"(r\d\d)_", os.path.basename(cur_path)).group(1)
    prnd = re.search(r"(r\d\d)_", os.path.basename(prev_path)).group(1) if prev_path else "-"
    r, pr = d["roofline"], p.get("roofline", {})
    rows = [("headline: iLEQG solves/s, N = 50, n = 12, m = 4, CE batch 1024 (`value`)", d["value"], p.get("value"), "M"),
            ("… ms per batch (`ms_per_step`)", d["ms_per_step"], p.get("ms_per_step"), "ms"),
            ("… dominant kernel's launch by HIP events (`roofline.avg_launch_ms`)", r.get("avg_launch_ms"), pr.get("avg_launch_ms"), "ms"),
            ("`roofline.frac` (SURVEY §8d contract: algorithmic bytes ÷ time ÷ 8 TB/s)", r.get("frac"), pr.get("frac"), "frac"),
            ("`roofline.frac_of_bound` = `fp64_frac` (5.68 Mflop per solve ÷ time ÷ 78.6 TFLOP/s)", r.get("frac_of_bound"), pr.get("frac_of_bound"), "frac"),
            ("`roofline.hbm_real_frac` (counter traffic ÷ time ÷ peak)", r.get("hbm_real_frac"), pr.get("hbm_real_frac"), "frac"),
            ("`roofline.issue_frac` (SQ counters)", r.get("issue_frac"), pr.get("issue_frac"), "frac"),
            ("contract leg: tiles materialised, `initialize!` per sample (`contract_solves_per_s`)", d.get("contract_solves_per_s"), p.get("contract_solves_per_s"), "M"),
            ("… its `frac` (`contract_frac`)", r.get("contract_frac"), pr.get("contract_frac"), "frac"),
            ("shard of 512 samples (`shard512_ms`)", d.get("shard512_ms"), p.get("shard512_ms"), "ms"),
            ("shard of 256 samples (`shard256_ms`)", d.get("shard256_ms"), p.get("shard256_ms"), "ms"),
            ("shard of 128 samples (`shard128_ms`)", d.get("shard128_ms"), p.get("shard128_ms"), "ms"),
            ("handle of width 8, 1024 samples (`e8_solves_per_s`; round 6: the sequential rule)", d.get("e8_solves_per_s"), p.get("e8_solves_per_s"), "M"),
            ("… forced to its width (`e8_forced_solves_per_s`)", d.get("e8_forced_solves_per_s"), p.get("e8_solves_per_s"), "M"),
            ("handle of width 8, shard of 128 (`e8_shard128_ms`)", d.get("e8_shard128_ms"), p.get("e8_shard128_ms"), "ms"),
            ("… forced (`e8_forced_shard128_ms`)", d.get("e8_forced_shard128_ms"), p.get("e8_shard128_ms"), "ms"),
            ("cubic drift κ = 0.05, 1024 samples (`nonlinear_solves_per_s`)", d.get("nonlinear_solves_per_s"), p.get("nonlinear_solves_per_s"), "M"),
            ("2048 / 4096 / 8192 samples", None, None, None),
            ("… `batch2048_solves_per_s`", d.get("batch2048_solves_per_s"), p.get("batch2048_solves_per_s"), "M"),
            ("… `batch4096_solves_per_s`", d.get("batch4096_solves_per_s"), p.get("batch4096_solves_per_s"), "M"),
            ("… `batch8192_solves_per_s`", d.get("batch8192_solves_per_s"), p.get("batch8192_solves_per_s"), "M"),
            ("`compute_cost` with host arrays (`compute_cost_host_ms`)", d.get("compute_cost_host_ms"), p.get("compute_cost_host_ms"), "ms"),
            ("one `rat_ce_solve` (`ce_solve_ms`)", d.get("ce_solve_ms"), p.get("ce_solve_ms"), "ms"),
            ("one Nelder–Mead `solve!` (`nm_ms_per_solve`)", d.get("nm_ms_per_solve"), p.get("nm_ms_per_solve"), "ms"),
            ("general size 16 × 4 (`wide_16x4_solves_per_s`)", d.get("wide_16x4_solves_per_s"), p.get("wide_16x4_solves_per_s"), "k"),
            ("general size 24 × 8 (`wide_24x8_solves_per_s`)", d.get("wide_24x8_solves_per_s"), p.get("wide_24x8_solves_per_s"), "k"),
            ("general size 32 × 32 (`wide_32x32_solves_per_s`)", d.get("wide_32x32_solves_per_s"), p.get("wide_32x32_solves_per_s"), "k"),
            ("PETS, 10 k trajectories (`pets_traj_per_s`)", d.get("pets_traj_per_s"), p.get("pets_traj_per_s"), "M"),
            ("PETS, 10⁶ trajectories (`pets_1m_traj_per_s`)", d.get("pets_1m_traj_per_s"), p.get("pets_1m_traj_per_s"), "M"),
            ("one PETS `solve!` 5 × 100 × 100 (`pets_solve_ms`)", d.get("pets_solve_ms"), p.get("pets_solve_ms"), "ms"),
            ("CPU baseline: C oracle, all host threads (`cpu_baseline.value`, solves/s)", (d.get("cpu_baseline") or {}).get("value"), (p.get("cpu_baseline") or {}).get("value"), "k")]
    out = ["<!-- BENCH:BEGIN -->",
           f"**Numbers, one MI355X** (generated by `tools/bench_brief.py --update-design` from `{os.path.relpath(cur_path, ROOT)}`; previous round: "
           f"`{os.path.relpath(prev_path, ROOT) if prev_path else '-'}`; boxes differ by ±2 %).\n",
           f"| quantity (key of the bench line) | {rnd} | {prnd} |", "|---|---|---|"]
    for name, a, b, kind in rows:
        if kind is None:
            continue
        out.append(f"| {name} | {fmt(a, kind)} | {fmt(b, kind)} |")
    ka = [("fused", "solve_fused_kernel", "`solve_fused_kernel`, 1024 samples"), ("contract", "solve_fused_kernel", "… with tile records (contract leg)"),
          ("block128", "solve_block_psw_kernel", "`solve_block_psw_kernel`, 128 samples (two workgroups per sample)"),
          ("solo128", "solve_block_psw_kernel", "… one workgroup per sample (`psw_duo = 0`)"),
          ("block256", "solve_block_psw_kernel", "`solve_block_psw_kernel`, 256 samples"), ("block512", "solve_block_kernel", "`solve_block_kernel`, 512 samples"),
          ("fused_4096", "solve_fused_kernel", "`solve_fused_kernel` two samples per SIMD, 4096 samples")]
    lines = []
    for cfg, pat, name in ka:
        v = kernel_avg_us(rnd, cfg, pat)
        if v is not None:
            lines.append(f"| {name} | {v:.1f} µs | `profiles/{rnd}_kernel_stats_{cfg}.csv` |")
    if lines:
        out += ["", "rocprofv3 `--kernel-trace --stats` averages of the same kernels (another box, under the profiler):\n", "| kernel | average launch | file |", "|---|---|---|"] + lines
    cfgd = d.get("config", {})
    out += ["", f"Workload statistics of the timed batch: feasible fraction {cfgd.get('feasible_fraction')}, mean iterations {cfgd.get('mean_iters')}, mean line-search "
            f"evaluations {cfgd.get('mean_ls_evals')}; path `{cfgd.get('path')}`.", "<!-- BENCH:END -->"]
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--update-design":
        cur = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_bench_default.json")
        prev = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r05_bench_default.json")
        blk = design_block(cur, prev)
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        a, b = s.index("<!-- BENCH:BEGIN -->"), s.index("<!-- BENCH:END -->") + len("<!-- BENCH:END -->")
        open(path, "w").write(s[:a] + blk + s[b:])
        print(blk)
    else:
        brief(sys.argv[1])
