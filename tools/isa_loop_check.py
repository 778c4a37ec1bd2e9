"""Spill traffic inside the time loops of a kernel, from hipcc's device assembly ("no spill traffic inside a time loop").
   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DRAT_PART=<bit> -S --cuda-device-only -o part.s kernels.hip
   python tools/isa_loop_check.py part.s <substring of the mangled kernel name> [...]
A "time loop" is an innermost loop (a label that a later branch jumps back to, with no such loop inside) that contains f64 MFMAs.  Spill
registers are the VGPRs some v_writelane of the kernel targets (SGPR spills live in their lanes); a v_readlane FROM one of them is an SGPR
reload (the pivot-block v_readlane of the elimination reads data registers and is not counted).  Reported per loop: instructions, MFMAs,
scratch loads / stores (VGPR spills), v_writelane (SGPR spill stores), SGPR reloads."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pats = sys.argv[2:]
starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
for pat in pats:
    for s in starts:
        name = src[s].split(":")[0]
        if pat not in name:
            continue
        e = next(i for i in range(s, len(src)) if "s_endpgm" in src[i])
        body = src[s:e + 1]
        spill_regs = set()
        for l in body:
            m = re.match(r"\s*v_writelane_b32 (v\d+),", l)
            if m:
                spill_regs.add(m.group(1))
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        print(f"## `{name}`: {len(body)} lines, {len(spill_regs)} SGPR-spill VGPRs {sorted(spill_regs)}, "
              f"{sum('scratch_' in l for l in body)} scratch instructions, {sum('v_writelane' in l for l in body)} v_writelane in the whole kernel\n")
        print("| loop (lines) | instructions | f64 MFMAs | scratch ld/st | v_writelane | SGPR reloads (v_readlane from a spill VGPR) |")
        print("|---|---|---|---|---|---|")
        tot = [0, 0, 0]
        for a, b in sorted(inner):
            seg = [l for l in body[a:b + 1] if re.match(r"^\s+[a-z]", l) and not l.lstrip().startswith(";")]
            nm = sum("v_mfma_f64" in l for l in seg)
            if nm == 0:
                continue
            sc = sum("scratch_" in l for l in seg)
            wl = sum("v_writelane" in l for l in seg)
            rl = sum(1 for l in seg for m in [re.match(r"\s*v_readlane_b32 s\d+, (v\d+),", l)] if m and m.group(1) in spill_regs)
            tot = [tot[0] + sc, tot[1] + wl, tot[2] + rl]
            print(f"| {a}-{b} | {len(seg)} | {nm} | {sc} | {wl} | {rl} |")
        print(f"\nTime loops in total: {tot[0]} scratch instructions, {tot[1]} v_writelane, {tot[2]} SGPR reloads.\n")
