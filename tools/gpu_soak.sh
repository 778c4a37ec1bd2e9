#!/bin/bash
# the randomised parity soaks and the path stress runs (profiles/r04_soak.md); knobs: SOAK_SEED0, SOAK_N, STRESS_S
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
SOAK_SEED0=${SOAK_SEED0:-0} SOAK_N=${SOAK_N:-3000} timeout 3000 python tools/soak_parity.py > $O/soak_parity.log 2>&1; tail -4 $O/soak_parity.log
SOAK_N=${SOAK_PL_N:-400} timeout 1500 python tools/soak_parity_powerlaw.py > $O/soak_powerlaw.log 2>&1; tail -2 $O/soak_powerlaw.log
SOAK_WIDE=1 SOAK_N=${SOAK_WIDE_N:-300} timeout 1500 python tools/soak_parity.py > $O/soak_wide.log 2>&1; tail -2 $O/soak_wide.log
STRESS_S=${STRESS_S:-150} timeout 1500 python tools/stress_paths.py > $O/stress_paths.log 2>&1; tail -2 $O/stress_paths.log
timeout 900 python tools/stress_block.py > $O/stress_block.log 2>&1; tail -2 $O/stress_block.log
