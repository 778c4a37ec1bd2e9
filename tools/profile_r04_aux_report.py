"""profiles/<round>_aux_kernels.md from the passes of tools/profile_<round>_aux.sh (gpurun_out/<round>/aux; usage: profile_r04_aux_report.py [round], default r04): duration, HBM traffic and SQ counters of
the kernels outside the headline configuration."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
O = os.path.join(ROOT, "gpurun_out", RND, "aux")
HBM_PEAK, CLK = 8.0e12, 2.4e9


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


md = [f"# {RND}: counter evidence for the kernels outside the headline (tools/profile_{RND}_aux.sh; one MI355X, ROCm 7.2)\n",
      "Runs: `tools/pets_bench.py` (BASELINE config 5: 100 x 100, 1000 x 100 and 1000 x 1000 control samples x noisy rollouts, N = 30),",
      "`tools/aux_wide_run.py` (general sizes: CE batch 1024 at n x m = 16 x 4 and 32 x 32, N = 50) and `tools/aux_nm_run.py` (BASELINE config 4:",
      "fresh Nelder-Mead solves: two iterations' worth of vertices per batch, the final solve read out of the last batch).  Each counter set is its own rocprofv3 run.",
      "HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 tallies 128-B read requests at 64 B); `issue` = (4 x SQ_ACTIVE_INST_VALU +",
      "SQ_VALU_MFMA_BUSY_CYCLES) / (4 x SQ_WAVE_CYCLES), the share of the resident waves' lifetime in which the FP64 datapath is issuing for them.\n",
      "| run | kernel | launches | avg (us) | max (us) | HBM traffic per launch, largest (MB) | HBM GB/s at the largest launch | of 8 TB/s | waves (largest) | VALU / MFMA / LDS / SALU / VMEM instr. per wave (k) | VALU issuing | issue | LDS active | LDS bank-conflict cycles / LDS active | s_waitcnt | parked |",
      "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for name in ("pets", "wide", "nm"):
    st = glob.glob(os.path.join(O, f"kt_{name}", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    import shutil
    shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{RND}_kernel_stats_aux_{name}.csv"))
    f, w = counters(f"fetch_{name}"), counters(f"write_{name}")
    s = collections.defaultdict(dict)
    for d in ("sq1", "sq2", "sq3"):
        for k, c in counters(f"{d}_{name}").items():
            for cn, v in c.items():
                s[k][cn] = v
    for row in csv.DictReader(open(st[0])):
        k = row["Name"]
        if not any(p in k for p in ("pets_", "wide_", "solve_block", "solve_fused", "rollin", "sweep")):
            continue
        avg, mx, calls = float(row["AverageNs"]) / 1e3, float(row["MaxNs"]) / 1e3, int(row["Calls"])
        t = (2.0 * max(f[k]["FETCH_SIZE"]) + max(w[k]["WRITE_SIZE"])) * 1024 if k in f and k in w else None
        c = s.get(k, {})
        def big(cn):            # the launch with the most waves (the largest problem of the run)
            if cn not in c or "SQ_WAVES" not in c:
                return float("nan")
            return max(c[cn])
        nw, wc = big("SQ_WAVES"), big("SQ_WAVE_CYCLES")
        g = big
        md.append(f"| {name} | `{k[:60]}` | {calls} | {avg:.1f} | {mx:.1f} | " + (f"{t / 1e6:.1f} | {t / (mx * 1e-6) / 1e9:.0f} | {t / (mx * 1e-6) / HBM_PEAK * 100:.1f} % | " if t else "- | - | - | ")
                  + f"{nw:.0f} | {g('SQ_INSTS_VALU') / nw / 1e3:.2f} / {g('SQ_INSTS_MFMA') / nw / 1e3:.2f} / {g('SQ_INSTS_LDS') / nw / 1e3:.2f} / "
                  f"{g('SQ_INSTS_SALU') / nw / 1e3:.2f} / {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / nw / 1e3:.2f} | "
                  f"{g('SQ_ACTIVE_INST_VALU') / wc * 100:.0f} % | "
                  f"{(g('SQ_ACTIVE_INST_VALU') * 4 + g('SQ_VALU_MFMA_BUSY_CYCLES')) / (wc * 4) * 100:.0f} % | {g('SQ_ACTIVE_INST_LDS') / wc * 100:.0f} % | "
                  f"{g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1) * 100:.0f} % | {g('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | {g('SQ_WAIT_ANY') / wc * 100:.0f} % |")
    log = os.path.join(O, f"{name}.log")
    if os.path.exists(log):
        ln = [l.strip() for l in open(log) if l.startswith("{") or l.startswith("[")]
        if ln:
            md.append(f"| {name}: the run's own line | `{ln[-1][:600]}` |")
open(os.path.join(ROOT, "profiles", f"{RND}_aux_kernels.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
