"""profiles/ of a round from the rocprofv3 passes of tools/profile_r04.sh (gpurun_out/<round>/prof; usage: profile_report.py [round]): kernel-trace stats (csv, copied), HBM
traffic per launch of every solver kernel (profiles/traffic.json, stamped with the kernel-source hash bench.py checks) and an SQ-counter
digest per kernel.  Configurations that run several kernels per batch (the round-based speculative path) get one row per kernel and a
per-batch total."""
import collections, csv, glob, json, math, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
O = os.path.join(ROOT, "gpurun_out", RND, "prof")
# (seq128 before block128, paired_4096 before fused_4096: both map to one key of profiles/traffic.json, the default kernel keeps it)
CFG = {"fused": (1, 1024), "contract": (1, 1024), "seq128": (1, 128), "solo128": (1, 128), "block128": (1, 128), "block256": (1, 256), "block512": (1, 512),
       "e8_1024": (8, 1024), "e8f_1024": (8, 1024), "e8_128": (8, 128), "paired_4096": (1, 4096), "fused_4096": (1, 4096)}
# kernel-name pattern -> the key bench.py uses (traffic_for(f"{kind}_E{E}_B{B}")); first match wins
KINDS = [("solve_fused_kernel", "solve_fused"), ("solve_block_psw_kernel", "solve_block"), ("solve_block_kernel", "solve_block"), ("sweep_dual_kernel", "sweep_dual"), ("sweep_cand0_kernel", "sweep_cand0"),
         ("rollin_multi_kernel", "rollout_multi"), ("rollin_stage_kernel", "rollout"), ("rollin_kernel", "rollout_init"),
         ("copy_initial_kernel", "copy_initial"),
         ("sweep_kernel<false", "sweep_eval"), ("sweep_kernel<true", "sweep_gain"), ("materialize_kernel", "materialize"),
         ("ls_select_kernel", "select")]
STEPS = 10 + 2 + 5            # launches of a once-per-batch kernel in one bench.py run of profile_r03.sh (timed + warm-up + profiled pass), + conditioning


def kind_of(name):
    if "solve_fused_kernel<" in name and re.search(r", true>\(", name):      # last template argument MAT: tile records materialised
        return "solve_fused_mat"
    for pat, k in KINDS:
        if pat in name:
            return k
    return None


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


def by_kind(per, counter, how=max):
    out = {}
    for kname, c in per.items():
        k = kind_of(kname)
        if k and counter in c:
            # several instantiations may share a kind (e.g. two sweep_dual variants): keep the one with most launches, then `how` over launches
            if k not in out or len(c[counter]) > out[k][1]:
                out[k] = (how(c[counter]), len(c[counter]))
    return {k: v[0] for k, v in out.items()}


traffic = {"kernels_sha": bench.kernel_source_hash(), "round": RND, "sq": {}}
md = [f"# {RND} rocprofv3 summaries (tools/profile_{RND}.sh; one MI355X, ROCm 7.2)\n",
      "Commands: `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / two `--pmc SQ_*` passes (each its own run, no trace",
      "domains with counters) around `python3 bench.py --steps 10 --warmup 2 --no-cpu --no-second [--batch B] [--spec-eps E]`.\n",
      "## Kernel duration (kernel-trace stats) and HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-B read requests at 64 B)\n",
      "| configuration | kernel | launches | avg duration (us) | HBM traffic per launch (MB) | algorithmic bytes per launch (MB) |", "|---|---|---|---|---|---|"]
a = bench.algo_bytes()
for name, (E, B) in CFG.items():
    st = glob.glob(os.path.join(O, f"kt_{name}", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{RND}_kernel_stats_{name}.csv"))
    log = os.path.join(O, f"bench_{name}_under_rocprof.log")
    if os.path.exists(log):
        ln = [l for l in open(log) if l.startswith("{")]
        if ln:
            open(os.path.join(ROOT, "profiles", f"{RND}_bench_{name}_under_rocprof.json"), "w").write(ln[-1])
    f, w = counters(f"fetch_{name}"), counters(f"write_{name}")
    fk, wk = by_kind(f, "FETCH_SIZE"), by_kind(w, "WRITE_SIZE")
    rows, per_batch, per_batch_t = {}, 0.0, 0.0
    for row in csv.DictReader(open(st[0])):
        k = kind_of(row["Name"])
        if not k:
            continue
        if k not in rows or int(row["Calls"]) > rows[k][1]:
            rows[k] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]), row["Name"])
    runs = None
    for k, (avg, calls, kname) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        t = (2.0 * fk[k] + wk[k]) * 1024 if k in fk and k in wk else None
        if t is not None:
            traffic[f"{k}_E{E}_B{B}"] = t
        alg = None
        if k in ("solve_fused", "solve_block", "solve_fused_mat"):
            # (a handle of width 8 under the default policy runs the sequential rule: the E = 1 solve's bytes)
            alg = B * (1537256 if (E == 1 or name == "e8_1024") else 368312 + 2 * 188864 + 2 * E * 395608)
        elif k == "sweep_eval":
            alg = a["sweep_eval"] * B * (E - 1 if ("sweep_dual" in rows or "sweep_cand0" in rows) else E)
        elif k in ("sweep_dual", "sweep_cand0"):
            alg = (a["sweep_eval"] + a["sweep_gain"]) * B
        elif k in ("rollout", "rollout_multi"):
            alg = (a["rollout_candidate"] + a["linearise"]) * B * E
        elif k == "rollout_init":
            alg = (a["rollout_init"] + a["linearise"]) * (1 if "copy_initial" in rows else B)
        elif k == "copy_initial":
            alg = (a["rollout_init"] + a["linearise"]) * B
        md.append(f"| {name}: B = {B}, E = {E} | `{kname[:70]}` | {calls} | {avg:.1f} | {t / 1e6:.1f} | {alg / 1e6 if alg else float('nan'):.1f} |"
                  if t is not None else f"| {name}: B = {B}, E = {E} | `{kname[:70]}` | {calls} | {avg:.1f} | - | {alg / 1e6 if alg else float('nan'):.1f} |")
        if E > 1 and k not in ("solve_block",) and "solve_fused" not in rows:
            # batches in the run: one copy of the shared initialize! trajectory per batch (the rollin_kernel launch is then the ONE
            # rollout per rat_set_initial); before round 3's sharing: one initial rollout per batch
            runs = runs or rows.get("copy_initial", rows.get("rollout_init", rows.get("sweep_dual")))[1]
            if not (k == "rollout_init" and "copy_initial" in rows):
                per_batch += avg * calls
                per_batch_t += (t or 0.0) * calls
    if E > 1 and runs and "solve_block" not in rows and "solve_fused" not in rows:
        batches = runs
        traffic[f"batch_E{E}_B{B}"] = per_batch_t / batches
        md.append(f"| {name}: per batch (all kernels) | | {batches} batches | {per_batch / batches:.1f} | {per_batch_t / batches / 1e6:.1f} | "
                  f"{B * (368312 + 2 * 188864 + 2 * E * 395608) / 1e6:.1f} |")

md += ["", "## SQ counters per kernel, per launch (max over launches)\n",
       "SQ cycle counters tick every 4 clocks.  `issue` = (4 x SQ_ACTIVE_INST_VALU + SQ_VALU_MFMA_BUSY_CYCLES) / (4 x SQ_WAVE_CYCLES): share of the waves'",
       "lifetime in which the FP64 datapath is issuing vector or matrix work (they do not overlap on gfx950: profiles/r01_ubench_fp64_pipe.md).\n",
       "| configuration | kernel | waves | wave-clocks per wave (k) | VALU / MFMA / LDS / SALU / VMEM instr. per wave (k) | VALU issuing | MFMA busy | issue | s_waitcnt (WAIT_INST_ANY) | parked (WAIT_ANY) |",
       "|---|---|---|---|---|---|---|---|---|---|"]
for name, (E, B) in CFG.items():
    c = collections.defaultdict(dict)
    for d in (f"sq1_{name}", f"sq2_{name}"):
        per = counters(d)
        for cn in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD",
                   "SQ_INSTS_VMEM_WR", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            for k, v in by_kind(per, cn).items():
                c[k][cn] = v
    for k, cc in sorted(c.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if "SQ_WAVE_CYCLES" not in cc or cc.get("SQ_WAVES", 0) < 64:
            continue
        wc, nw = cc["SQ_WAVE_CYCLES"], cc["SQ_WAVES"]
        g = lambda n: cc.get(n, float("nan"))
        issue = (g('SQ_ACTIVE_INST_VALU') * 4 + g('SQ_VALU_MFMA_BUSY_CYCLES')) / (wc * 4)
        traffic["sq"][f"{k}_E{E}_B{B}"] = {"issue_frac": issue, "valu_frac": g('SQ_ACTIVE_INST_VALU') / wc, "mfma_frac": g('SQ_VALU_MFMA_BUSY_CYCLES') / (wc * 4),
                                          "wait_inst_frac": g('SQ_WAIT_INST_ANY') / wc, "parked_frac": g('SQ_WAIT_ANY') / wc, "waves": nw,
                                          "wave_clocks_per_wave": wc * 4 / nw, "valu_per_wave": g('SQ_INSTS_VALU') / nw, "mfma_per_wave": g('SQ_INSTS_MFMA') / nw}
        md.append(f"| {name} | {k} | {nw:.0f} | {wc * 4 / nw / 1e3:.1f} | {g('SQ_INSTS_VALU') / nw / 1e3:.2f} / {g('SQ_INSTS_MFMA') / nw / 1e3:.2f} / "
                  f"{g('SQ_INSTS_LDS') / nw / 1e3:.2f} / {g('SQ_INSTS_SALU') / nw / 1e3:.2f} / {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / nw / 1e3:.2f} | "
                  f"{g('SQ_ACTIVE_INST_VALU') / wc * 100:.0f} % | {g('SQ_VALU_MFMA_BUSY_CYCLES') / (wc * 4) * 100:.0f} % | "
                  f"{(g('SQ_ACTIVE_INST_VALU') * 4 + g('SQ_VALU_MFMA_BUSY_CYCLES')) / (wc * 4) * 100:.0f} % | {g('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | {g('SQ_WAIT_ANY') / wc * 100:.0f} % |")
def _clean(o):            # (a counter missing from a pass reads NaN: null in the JSON)
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    return None if isinstance(o, float) and math.isnan(o) else o


json.dump(_clean(traffic), open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{RND}_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
