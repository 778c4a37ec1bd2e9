#!/bin/bash
# round-5 evidence on an MI355X (through gpurun): the time-parallel sweep's operator timings, phase timelines, solve timings, soak / stress
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
PSW_P=2,3,4,5,6,8 timeout 300 python tools/psweep_time.py 128 512 > $O/psweep_time.log 2>&1
timeout 300 python tools/psweep_phases.py 128 2 4 > $O/psweep_phases.log 2>&1
timeout 300 python tools/gpu_phases_bpsw.py 128 > $O/phases_bpsw.log 2>&1
PSW_REPS=50 timeout 300 python tools/psw_block_time.py 1 16 128 256 > $O/psw_block_time.log 2>&1
timeout 300 python tools/pets_solve_time.py > $O/pets_solve_time.log 2>&1
timeout 300 python tools/nm_time.py > $O/nm_time.log 2>&1
SOAK_E=1 SOAK_N=3000 timeout 900 python tools/soak_parity.py > $O/soak_psw.log 2>&1
SOAK_N=3000 timeout 900 python tools/soak_parity.py > $O/soak_default.log 2>&1
STRESS_PSW=1 STRESS_S=180 timeout 600 python tools/stress_block.py > $O/stress_psw.log 2>&1
STRESS_S=60 timeout 400 python tools/stress_block.py > $O/stress_block.log 2>&1
STRESS_S=60 timeout 400 python tools/stress_paths.py > $O/stress_paths.log 2>&1
tail -2 $O/soak_psw.log $O/soak_default.log $O/stress_psw.log $O/stress_block.log $O/stress_paths.log
