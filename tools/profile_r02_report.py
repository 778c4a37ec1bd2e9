"""profiles/ of round 2 from the rocprofv3 passes of tools/profile_r02.sh (gpurun_out/r02/prof): kernel-trace stats (csv, copied),
HBM traffic per launch (profiles/traffic.json, stamped with the kernel-source hash bench.py checks) and an SQ-counter digest."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
O = os.path.join(ROOT, "gpurun_out", "r02", "prof")
CFG = {"fused": ("solve_fused", 1, 1024), "block512": ("solve_block", 1, 512), "block128": ("solve_block", 1, 128), "e8": ("solve_block", 8, 128),
       "block1024": ("solve_block", 1, 1024), "occ2_4096": ("solve_fused", 1, 4096), "fused_4096": ("solve_fused_paired", 1, 4096)}


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


def solve_kernel(per):
    ks = [k for k in per if "solve_fused_kernel" in k or "solve_block_kernel" in k]
    return max(ks, key=lambda k: sum(len(v) for v in per[k].values())) if ks else None


traffic = {"kernels_sha": bench.kernel_source_hash(), "round": "r02"}
md = ["# r02 rocprofv3 summaries (tools/profile_r02.sh; one MI355X, ROCm 7.2)\n",
      "Commands: `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / two `--pmc SQ_*` passes (each its own run, no trace",
      "domains with counters) around `python3 bench.py --steps 10 --warmup 2 --no-cpu --no-second [--batch B] [--spec-eps E]`.\n",
      "## Kernel duration (kernel-trace stats) and HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-B read requests at 64 B)\n",
      "| configuration | kernel | launches | avg duration (us) | HBM traffic per launch (MB) | per solve (MB) | algorithmic (MB / solve) |", "|---|---|---|---|---|---|---|"]
for name, (kind, E, B) in CFG.items():
    st = glob.glob(os.path.join(O, f"kt_{name}", "**", "*kernel_stats.csv"), recursive=True)
    avg = calls = None
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"r02_kernel_stats_{name}.csv"))
        for row in csv.DictReader(open(st[0])):
            if "solve_fused_kernel" in row["Name"] or "solve_block_kernel" in row["Name"]:
                avg, calls = float(row["AverageNs"]) / 1e3, int(row["Calls"])
    log = os.path.join(O, f"bench_{name}_under_rocprof.log")
    if os.path.exists(log):
        ln = [l for l in open(log) if l.startswith("{")]
        if ln:
            open(os.path.join(ROOT, "profiles", f"r02_bench_{name}_under_rocprof.json"), "w").write(ln[-1])
    f, w = counters(f"fetch_{name}"), counters(f"write_{name}")
    kf, kw = solve_kernel(f), solve_kernel(w)
    t = None
    if kf and kw:
        t = (2.0 * max(f[kf]["FETCH_SIZE"]) + max(w[kw]["WRITE_SIZE"])) * 1024
        traffic[f"{kind}_E{E}_B{B}"] = t
    alg = (1537256 if E == 1 else 368312 + 2 * 188864 + 2 * E * 395608) / 1e6
    md.append(f"| {name}: B = {B}, E = {E} | `{kind}_kernel` | {calls} | {avg and round(avg, 1)} | {t and round(t / 1e6, 1)} | {t and round(t / B / 1e6, 3)} | {alg:.3f} |")
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)

md += ["", "## SQ counters of the solve kernel, per launch (max over launches) and per sample\n",
       "SQ cycle counters tick every 4 clocks.  `issue` = (4 x SQ_ACTIVE_INST_VALU + SQ_VALU_MFMA_BUSY_CYCLES) / (4 x SQ_WAVE_CYCLES): share of the waves'",
       "lifetime in which the FP64 datapath is issuing vector or matrix work (they do not overlap on gfx950: profiles/r01_ubench_fp64_pipe.md).\n",
       "| configuration | waves | wave-clocks per sample (M) | VALU / MFMA / LDS / SALU / VMEM instr. per sample (k) | VALU issuing | MFMA busy | issue | s_waitcnt (WAIT_INST_ANY) | parked (WAIT_ANY) |",
       "|---|---|---|---|---|---|---|---|---|"]
for name, (kind, E, B) in CFG.items():
    c = {}
    for d in (f"sq1_{name}", f"sq2_{name}"):
        per = counters(d)
        k = solve_kernel(per)
        if k:
            for cn, v in per[k].items():
                c[cn] = max(v)
    if "SQ_WAVE_CYCLES" not in c:
        continue
    wc = c["SQ_WAVE_CYCLES"]
    g = lambda n: c.get(n, float("nan"))
    md.append(f"| {name} | {g('SQ_WAVES'):.0f} | {wc * 4 / B / 1e6:.3f} | {g('SQ_INSTS_VALU') / B / 1e3:.1f} / {g('SQ_INSTS_MFMA') / B / 1e3:.1f} / "
              f"{g('SQ_INSTS_LDS') / B / 1e3:.1f} / {g('SQ_INSTS_SALU') / B / 1e3:.1f} / {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / B / 1e3:.1f} | "
              f"{g('SQ_ACTIVE_INST_VALU') / wc * 100:.0f} % | {g('SQ_VALU_MFMA_BUSY_CYCLES') / (wc * 4) * 100:.0f} % | "
              f"{(g('SQ_ACTIVE_INST_VALU') * 4 + g('SQ_VALU_MFMA_BUSY_CYCLES')) / (wc * 4) * 100:.0f} % | {g('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | {g('SQ_WAIT_ANY') / wc * 100:.0f} % |")
open(os.path.join(ROOT, "profiles", "r02_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
