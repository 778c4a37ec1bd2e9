#!/bin/bash
# Round-6 profiles on an MI355X (run through gpurun): kernel-trace stats, HBM traffic (separate --pmc passes, no trace domains in the
# same run) and SQ counters of
#   fused      B = 1024, E = 1  solve_fused_kernel: the headline
#   block128   B = 128,  E = 1  solve_block_psw_kernel, TWO workgroups (compute units) per sample (round 6): the per-GPU shard of the headline at 8 GPUs
#   solo128    B = 128,  E = 1  the same kernel with one workgroup per sample (switch psw_duo = 0: the round-5 schedule)
#   block256   B = 256,  E = 1  solve_block_psw_kernel, one workgroup per sample: the shard at 4 GPUs
#   block512   B = 512,  E = 1  solve_block_kernel: the per-GPU shard at 2 GPUs
#   e8_1024    B = 1024, handle of width 8 under the default policy (spec_eps an upper bound: the E = 1 kernel)
#   e8f_1024   B = 1024, E = 8 forced (spec_force = 1): round-based path without candidate tiles, pruned
#   fused_4096 B = 4096, E = 1  two samples per SIMD: the 256-register tile-free kernel (default beyond one sample per SIMD)
#   contract   B = 1024, E = 1  the headline kernel with tile records materialised and no shared initialize! (SURVEY 8d to the letter)
# Outputs under gpurun_out/r06/prof/; summarised into profiles/ by tools/profile_report.py r06.   usage: profile_r06.sh [config ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/prof; mkdir -p $O
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
want() { [ $# -eq 0 ] && return 0; for c in "$@"; do [ "$c" = "$CUR" ] && return 0; done; return 1; }
CUR=fused;     want "$@" && run fused --batch 1024
CUR=contract;  want "$@" && run contract --batch 1024 --debug materialize=1 --debug init_share=0
CUR=block512;  want "$@" && run block512 --batch 512
CUR=block128;  want "$@" && run block128 --batch 128
CUR=solo128;   want "$@" && run solo128 --batch 128 --debug psw_duo=0
CUR=block256;  want "$@" && run block256 --batch 256
CUR=e8_1024;   want "$@" && run e8_1024 --batch 1024 --spec-eps 8
CUR=e8f_1024;  want "$@" && run e8f_1024 --batch 1024 --spec-eps 8 --debug spec_force=1
CUR=fused_4096; want "$@" && run fused_4096 --batch 4096
find $O -name "*.csv" | wc -l
