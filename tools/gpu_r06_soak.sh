#!/bin/bash
# round-6 regression + soaks / race hunt of the two-workgroups-per-sample kernel (through gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/soak; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
STRESS_PSW=1 STRESS_S=${STRESS_S:-150} timeout 600 python tools/stress_block.py > $O/stress_psw.log 2>&1
SOAK_E=1 SOAK_N=${SOAK_N:-3000} timeout 900 python tools/soak_parity.py > $O/soak_psw.log 2>&1
tail -n 3 $O/pytest_gpu.log; tail -n 2 $O/stress_psw.log $O/soak_psw.log
