#!/bin/bash
# Round-5 counter evidence for the kernels outside the headline: PETS (BASELINE config 5: pets_rollout16(s)_kernel / pets_mean_kernel),
# the general-size solve (wide_solve_kernel, n = 16 and n = m = 32) and Nelder-Mead's simplex batches (BASELINE config 4: solve_block_kernel
# at 80..158 samples: two Nelder-Mead iterations per batch).  kernel-trace stats, FETCH_SIZE / WRITE_SIZE and two SQ passes, each its own run (no trace domains with counters).
# Outputs under gpurun_out/r05/aux/; summarised into profiles/r05_aux_kernels.md by tools/profile_r05_aux_report.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/aux; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
run() {  # name, script args
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o runc --output-format csv -- python3 "$@" > $O/${name}.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$name -o runc --output-format csv -- python3 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/write_$name -o runc --output-format csv -- python3 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/sq1_$name -o runc --output-format csv -- python3 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq2_$name -o runc --output-format csv -- python3 "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d $O/sq3_$name -o runc --output-format csv -- python3 "$@" > /dev/null 2>&1
}
run pets tools/pets_bench.py
run wide tools/aux_wide_run.py
run nm tools/aux_nm_run.py
find $O -name "*.csv" | wc -l
