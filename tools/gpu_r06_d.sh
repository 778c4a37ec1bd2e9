#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 120 tools/ubench/elim_variants > $O/elim_variants.txt 2>&1; cat $O/elim_variants.txt
timeout 300 python tools/gpu_phases_duo.py 128 > $O/phases_duo.txt 2>&1; tail -4 $O/phases_duo.txt
timeout 300 python tools/gpu_phases_bpsw.py 256 > $O/phases_bpsw256.txt 2>&1; tail -2 $O/phases_bpsw256.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python tools/bench_brief.py $O/bench_default.json 2>&1 | head -60
