#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
B="--steps 10 --warmup 2 --no-cpu --no-second --condition-seconds 0.05"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_e8_1024 -o runc --output-format csv -- python3 bench.py $B --batch 1024 --spec-eps 8 > $O/bench_e8_1024_under_rocprof.log 2>&1
f=$(find $O/kt_e8_1024 -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -20
