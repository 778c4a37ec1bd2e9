#!/bin/bash
# round-5 soaks and race hunts on the final kernels (through gpurun); results summarised in profiles/r05_soak.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/final; mkdir -p $O
cd $R
SOAK_E=1 SOAK_N=3000 timeout 900 python tools/soak_parity.py > $O/soak_psw.log 2>&1
SOAK_N=3000 timeout 900 python tools/soak_parity.py > $O/soak_default.log 2>&1
SOAK_WIDE=16 SOAK_N=1500 timeout 600 python tools/soak_parity.py > $O/soak_wide16.log 2>&1
SOAK_WIDE=1 SOAK_N=600 timeout 600 python tools/soak_parity.py > $O/soak_wide.log 2>&1
STRESS_PSW=1 STRESS_S=150 timeout 600 python tools/stress_block.py > $O/stress_psw.log 2>&1
STRESS_S=60 timeout 400 python tools/stress_block.py > $O/stress_block.log 2>&1
STRESS_S=180 timeout 400 python tools/stress_paths.py > $O/stress_paths.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -n 1 $O/soak_psw.log $O/soak_default.log $O/soak_wide16.log $O/soak_wide.log $O/stress_psw.log $O/stress_block.log $O/stress_paths.log
grep "passed\|failed" $O/pytest_gpu.log
