#!/bin/bash
# full GPU suite + the randomised parity soak and the path stress runs (profiles/r03_soak.md)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
SOAK_N=${SOAK_N:-2000} timeout 1500 python tools/soak_parity.py > $O/soak_parity.log 2>&1; tail -4 $O/soak_parity.log
STRESS_S=150 timeout 900 python tools/stress_paths.py > $O/stress_paths.log 2>&1; tail -2 $O/stress_paths.log
timeout 600 python tools/stress_block.py > $O/stress_block.log 2>&1; tail -2 $O/stress_block.log
