#!/bin/bash
# round-6 final evidence on one MI355X (through gpurun): profiles of every benchmarked configuration, the default bench line, the timeline of
# the latency kernel, the PETS solve; summarised afterwards by tools/profile_report.py r06 and tools/bench_brief.py --update-design
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
bash tools/profile_r06.sh > $O/profile.log 2>&1; tail -n 1 $O/profile.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
WAVE1=1 timeout 300 python tools/gpu_phases_duo.py 128 > $O/phases_duo.txt 2>&1; tail -n 12 $O/phases_duo.txt | cut -c1-260
timeout 300 python tools/prl_time.py 1 16 128 > $O/prl_time.txt 2>&1; cat $O/prl_time.txt
timeout 300 python tools/duo_time.py 1 16 64 128 > $O/duo_time.txt 2>&1; cat $O/duo_time.txt
timeout 300 python tools/pets_solve_time.py > $O/pets_solve_time.txt 2>&1; cat $O/pets_solve_time.txt
