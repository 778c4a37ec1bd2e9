"""Writes profiles/r01_pmc_sq_E1.md from the SQ counter passes of tools/profile_sq.sh (gpurun_out/r01b/pmc_sq{1,2})."""
import collections, csv, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r01b")
rows, fused = [], {}
for d in ("pmc_sq1", "pmc_sq2"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f"{O}/{d}/runc_counter_collection.csv")):
        per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in per:
        if "solve_fused" in k or "sweep_kernel<false, false, false, true>" in k or "rollin_kernel<1, 1" in k:
            for c, v in per[k].items():
                rows.append((k[:60], c, max(v)))
                if "solve_fused" in k:
                    fused[c] = max(v) / 1024
rows.sort()
md = ("# r01 SQ counters per launch (E = 1, B = 1024 waves of 64 lanes; max over launches), tools/profile_sq.sh\n\n"
      "| kernel | counter | per launch | per wave |\n|---|---|---|---|\n")
for k, c, v in rows:
    md += f"| `{k}` | {c} | {v:.0f} | {v/1024:.0f} |\n"
wc = fused["SQ_WAVE_CYCLES"]
md += f"""
## Reading (`solve_fused_kernel`, one wave = one complete 2-iteration solve; SQ cycle counters tick every 4 clocks)

* SQ_WAVE_CYCLES {wc:,.0f} x 4 = {wc*4/1e6:.2f} M clocks per solve (tools/gpu_phases.py agrees).
* {fused['SQ_INSTS_VALU']/1e3:.1f} k VALU + {fused['SQ_INSTS_MFMA']/1e3:.1f} k MFMA + {fused['SQ_INSTS_LDS']/1e3:.1f} k LDS + {fused['SQ_INSTS_SALU']/1e3:.1f} k SALU + {(fused['SQ_INSTS_VMEM_RD']+fused['SQ_INSTS_VMEM_WR'])/1e3:.1f} k VMEM instructions per solve.
* VALU issuing {fused['SQ_ACTIVE_INST_VALU']/wc*100:.0f} % of the wave's life (SQ_ACTIVE_INST_VALU), any instruction issuing {fused['SQ_ACTIVE_INST_ANY']/wc*100:.0f} % (SQ_ACTIVE_INST_ANY),
  waiting on `s_waitcnt` {fused['SQ_WAIT_INST_ANY']/wc*100:.0f} % (SQ_WAIT_INST_ANY); the rest are dependency stalls of the in-order wave.
  MFMA pipe busy {fused['SQ_VALU_MFMA_BUSY_CYCLES']:,.0f} clocks = {fused['SQ_INSTS_MFMA']:,.0f} MFMAs x 64 = {fused['SQ_VALU_MFMA_BUSY_CYCLES']/(wc*4)*100:.0f} %.
* An f64 MFMA and vector instructions do not overlap on this hardware (profiles/r01_ubench_fp64_pipe.md: MFMA + VALU costs the sum of
  the two at every occupancy), so VALU issue + MFMA busy = {(fused['SQ_ACTIVE_INST_VALU']*4+fused['SQ_VALU_MFMA_BUSY_CYCLES'])/(wc*4)*100:.0f} % of the wave's life is a serial FP64-datapath floor.
  With one wave per SIMD (1024 samples on 1024 SIMDs) the rest are dependency stalls of the in-order wave (elimination of M) and the
  write-bound rollout phases: the work went into shortening that chain, pairing two recursions in one wave, removing every wait that
  is not a true dependency, and trimming vector/matrix instructions per step -- not into bandwidth.
"""
open(os.path.join(ROOT, "profiles", "r01_pmc_sq_E1.md"), "w").write(md)
print(md[-900:])
