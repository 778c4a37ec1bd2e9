#!/bin/bash
# Round-2 profiles on an MI355X (run through gpurun): kernel-trace stats, HBM traffic (separate --pmc passes, no trace domains in
# the same run) and SQ counters of the three single-launch solve configurations:
#   fused   B = 1024, E = 1  (solve_fused_kernel: the headline)        block512 / block128: B = 512 / 128, E = 1 (solve_block_kernel:
#   the per-GPU shards of strong scaling over 2 / 8 GPUs)              e8: B = 128, E = 8 (solve_block_kernel, config 3's shard at 8 GPUs)
#   occ2_4096 / fused_4096: B = 4096, E = 1 with two samples per SIMD (default beyond 1024 samples) / the paired kernel in generations
# Outputs under gpurun_out/r02/prof/; summarised into profiles/ by tools/profile_r02_report.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
B="--steps 10 --warmup 2 --no-cpu --no-second --condition-seconds 0.05"
run() {  # name, bench args
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o runc --output-format csv -- python3 bench.py $B "$@" > $O/bench_${name}_under_rocprof.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$name -o runc --output-format csv -- python3 bench.py $B "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write_$name -o runc --output-format csv -- python3 bench.py $B "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/sq1_$name -o runc --output-format csv -- python3 bench.py $B "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $O/sq2_$name -o runc --output-format csv -- python3 bench.py $B "$@" > /dev/null 2>&1
}
run fused --batch 1024
run block512 --batch 512
run block128 --batch 128
run e8 --batch 128 --spec-eps 8
RATILQR_BLOCK=1 run block1024 --batch 1024
run occ2_4096 --batch 4096
RATILQR_FUSED_OCC2=0 run fused_4096 --batch 4096
find $O -name "*.csv" | wc -l
tail -1 $O/bench_fused_under_rocprof.log | cut -c1-200
