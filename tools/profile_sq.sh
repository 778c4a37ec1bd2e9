# SQ instruction-mix counters of the default E = 1 run (own passes, --pmc only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/pmc_sq1 -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-second > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq2 -o runc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-second > $O/sq2.log 2>&1
ls $O/pmc_sq1 $O/pmc_sq2; tail -3 $O/sq2.log | cut -c1-200
