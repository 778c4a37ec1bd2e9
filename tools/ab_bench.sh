#!/bin/bash
# A/B of library variants under build/ab/*.so on ONE box: headline batch, interleaved repetitions (boxes differ by +-2 %)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
for v in build/ab/*.so; do
RATILQR_SO=$R/$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu --no-second "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'rep $rep', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
done; done
