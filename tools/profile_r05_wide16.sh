#!/bin/bash
# wide_solve_kernel at 16 x 4 alone (the register form of wide16.h) under rocprofv3: kernel-trace stats and two SQ counter passes, each its own
# run.  Outputs under gpurun_out/r05/w16/; summarised by tools/profile_r05_wide16_report.py into profiles/r05_wide16_counters.md.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/w16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o runc --output-format csv -- python3 tools/aux_wide_run.py 16x4 > $O/run.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/sq1 -o runc --output-format csv -- python3 tools/aux_wide_run.py 16x4 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq2 -o runc --output-format csv -- python3 tools/aux_wide_run.py 16x4 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o runc --output-format csv -- python3 tools/aux_wide_run.py 16x4 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o runc --output-format csv -- python3 tools/aux_wide_run.py 16x4 > /dev/null 2>&1
find $O -name "*.csv" | wc -l
