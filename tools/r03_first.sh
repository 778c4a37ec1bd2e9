#!/bin/bash
# round 3, first GPU pass: parity tests of the speculative paths with and without candidate tiles, E = 8 bench both ways
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="--steps 10 --warmup 3 --no-cpu --no-second"
timeout 300 python bench.py $B --batch 1024 --spec-eps 8 > $O/bench_e8_fly.json 2> $O/bench_e8_fly.err
RATILQR_FLY=0 timeout 300 python bench.py $B --batch 1024 --spec-eps 8 > $O/bench_e8_nofly.json 2> $O/bench_e8_nofly.err
for f in fly nofly; do python3 - <<PY
import json
d=json.loads(open("$O/bench_e8_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d["config"].get("mean_iters"))
PY
done
