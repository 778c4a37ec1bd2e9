"""profiles/r05_wide16_counters.md from the passes of tools/profile_r05_wide16.sh (gpurun_out/r05/w16): wide_solve_kernel at 16 x 4 alone."""
import collections, csv, glob, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r05", "w16")


def counters(d):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "wide_solve_kernel" in row["Kernel_Name"]:
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: max(v) for k, v in per.items()}


c = {}
for d in ("sq1", "sq2", "fetch", "write"):
    c.update(counters(d))
st = glob.glob(os.path.join(O, "kt", "**", "*kernel_stats.csv"), recursive=True)
avg = calls = None
if st:
    shutil.copy(st[0], os.path.join(ROOT, "profiles", "r05_kernel_stats_wide16.csv"))
    for row in csv.DictReader(open(st[0])):
        if "wide_solve_kernel" in row["Name"]:
            avg, calls = float(row["AverageNs"]) / 1e3, int(row["Calls"])
wc, nw = c["SQ_WAVE_CYCLES"], c["SQ_WAVES"]
g = lambda n: c.get(n, float("nan"))
md = ["# r05: SQ counters of `wide_solve_kernel` at 16 x 4 alone -- the register form of `wide16.h` (tools/profile_r05_wide16.sh; one MI355X)\n",
      "`tools/aux_wide_run.py 16x4` (CE batch 1024, N = 50, two iterations per sample) under `rocprofv3 --kernel-trace --stats` and four `--pmc` passes,",
      "each its own run.  Per launch of 1,024 waves, one per SIMD (round 4, the general LDS sweep at the same size: `profiles/r04_wide_sizes.md`).\n",
      "| quantity | r05 (`wide16.h`) | r04 (general sweep) |", "|---|---|---|",
      f"| kernel duration, average of {calls} launches (us) | {avg:.1f} | 5,070 |",
      f"| wave-clocks per wave (k) | {wc * 4 / nw / 1e3:.0f} | 11,900 |",
      f"| vector / matrix / LDS / scalar / memory instructions per wave (k) | {g('SQ_INSTS_VALU') / nw / 1e3:.1f} / {g('SQ_INSTS_MFMA') / nw / 1e3:.2f} / "
      f"{g('SQ_INSTS_LDS') / nw / 1e3:.1f} / {g('SQ_INSTS_SALU') / nw / 1e3:.1f} / {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / nw / 1e3:.1f} | 687 / 6.75 / 94 / 401 / 6.6 |",
      f"| vector ALU issuing (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) | {g('SQ_ACTIVE_INST_VALU') / wc * 100:.0f} % | 24 % |",
      f"| matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 4 / SQ_WAVE_CYCLES) | {g('SQ_VALU_MFMA_BUSY_CYCLES') / (wc * 4) * 100:.0f} % | 4 % |",
      f"| FP64 datapath issuing (the two above: they do not overlap on gfx950) | {(g('SQ_ACTIVE_INST_VALU') * 4 + g('SQ_VALU_MFMA_BUSY_CYCLES')) / (wc * 4) * 100:.0f} % | 28 % |",
      f"| any instruction issuing (SQ_ACTIVE_INST_ANY) | {g('SQ_ACTIVE_INST_ANY') / wc * 100:.0f} % | 46 % |",
      f"| parked (SQ_WAIT_ANY) | {g('SQ_WAIT_ANY') / wc * 100:.0f} % | 48 % |",
      f"| s_waitcnt (SQ_WAIT_INST_ANY) | {g('SQ_WAIT_INST_ANY') / wc * 100:.0f} % | 6 % |",
      f"| HBM traffic per launch, 2 x FETCH_SIZE + WRITE_SIZE (MB) | {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e6:.1f} | - |"]
open(os.path.join(ROOT, "profiles", "r05_wide16_counters.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
