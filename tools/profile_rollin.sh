# Where the closed-loop rollin kernel waits (round-based path, E = 1): store-side SQ counters (a TA_* / TCC_* pass hung the profiler on this pool: not collected)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rollin_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
export RATILQR_FUSED=0
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM -d $O/p1 -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-second > $O/p1.log 2>&1
ls $O/p1
