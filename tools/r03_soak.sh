#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
SOAK_N=${SOAK_N:-1500} timeout 1500 python tools/soak_parity.py > $O/soak_parity.log 2>&1; tail -5 $O/soak_parity.log
SOAK_N=300 timeout 600 python tools/soak_parity_powerlaw.py > $O/soak_powerlaw.log 2>&1; tail -3 $O/soak_powerlaw.log
timeout 900 python tools/stress_paths.py > $O/stress_paths.log 2>&1; tail -3 $O/stress_paths.log
