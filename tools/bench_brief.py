"""Prints the headline and the secondaries of a bench.py JSON line.   python tools/bench_brief.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 4), "path", d["config"].get("path"))
for k in ("contract_solves_per_s", "e8_solves_per_s", "e8_shard128_ms", "nonlinear_solves_per_s", "shard512_ms", "shard256_ms", "shard128_ms", "ce_solve_ms",
          "compute_cost_host_ms", "nm_ms_per_solve", "wide_16x4_solves_per_s", "wide_32x32_solves_per_s", "pets_traj_per_s", "pets_1m_traj_per_s", "pets_solve_ms",
          "steady_solves_per_s"):
    print("  ", k, d.get(k))
r = d["roofline"]
print("roofline:", {k: r.get(k) for k in ("bound", "frac", "frac_of_bound", "fp64_frac", "issue_frac", "hbm_real_frac", "contract_frac", "traffic", "avg_launch_ms")})
print("cpu_baseline:", d.get("cpu_baseline"))
