#!/bin/bash
# a longer soak of the final round-6 build on fresh seeds (through gpurun): problems 3000 .. 11999 of tools/soak_parity.py on the default paths and
# on the time-parallel latency kernel, 2,400 more at general sizes, and a ten-minute race hunt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/longsoak; mkdir -p $O
cd $R
SOAK_SEED0=3000 SOAK_N=9000 timeout 900 python tools/soak_parity.py > $O/default.log 2>&1
SOAK_E=1 SOAK_SEED0=3000 SOAK_N=9000 timeout 900 python tools/soak_parity.py > $O/e1.log 2>&1
SOAK_WIDE=1 SOAK_SEED0=800 SOAK_N=2400 timeout 900 python tools/soak_parity.py > $O/wide.log 2>&1
SOAK_WIDE=16 SOAK_SEED0=800 SOAK_N=2400 timeout 900 python tools/soak_parity.py > $O/wide16.log 2>&1
STRESS_PSW=1 STRESS_S=600 timeout 1200 python tools/stress_block.py > $O/stress_psw.log 2>&1
for f in default e1 wide wide16 stress_psw; do echo "== $f: $(grep -c MISMATCH $O/$f.log) mismatch lines; $(tail -n 1 $O/$f.log)"; done
grep -h MISMATCH $O/*.log | cut -c1-90
