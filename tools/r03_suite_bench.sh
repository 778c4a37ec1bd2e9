#!/bin/bash
# full GPU suite, then the headline and E = 8 bench lines (no CPU leg)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
B="--steps 20 --warmup 5 --no-cpu --no-second"
run() { name=$1; shift; env "$@" > $O/bench_$name.json 2> $O/bench_$name.err
python3 - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"]), round(d["ms_per_step"],4), {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items() if v}, d["config"].get("mean_iters"))
except Exception as e: print("$name FAILED", e)
PY
}
run e1_1024 timeout 300 python bench.py $B --batch 1024
run e1_1024_nowd RATILQR_WDIAG=0 timeout 300 python bench.py $B --batch 1024
run e1_128 timeout 300 python bench.py $B --batch 128
run e1_128_nowd RATILQR_WDIAG=0 timeout 300 python bench.py $B --batch 128
run e1_4096 timeout 300 python bench.py $B --batch 4096
run e8_1024 timeout 300 python bench.py $B --batch 1024 --spec-eps 8
run e8_1024_nowd RATILQR_WDIAG=0 timeout 300 python bench.py $B --batch 1024 --spec-eps 8
run e8_128 timeout 300 python bench.py $B --batch 128 --spec-eps 8
