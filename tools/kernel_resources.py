"""Register / LDS / occupancy table of the solve kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
  hipcc ... -Rpass-analysis=kernel-resource-usage -c kernels.hip 2> remarks.txt ; python tools/kernel_resources.py remarks.txt [filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:] or ["solve_fused", "solve_block", "sweep_dual"]
print("| kernel | VGPR | AGPR | SGPR | scratch B/lane | SGPR spills | VGPR spills | LDS B/block | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|---|")
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    if not any(f in name for f in filt):           # (filters match the demangled name, template arguments included)
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    cols = [g(k) for k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]",
                           "Occupancy [waves/SIMD]")]
    print(f"| `{name}` | " + " | ".join(cols) + " |")
