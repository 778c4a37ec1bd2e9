#!/bin/bash
# quick regression on an MI355X (through gpurun): the parity-critical GPU tests + the main bench points
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_ce.py -m gpu -x -q > $O/pytest_quick.log 2>&1; echo "pytest rc=$?" >> $O/pytest_quick.log
tail -3 $O/pytest_quick.log
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
run e1_128 timeout 300 python bench.py $B --batch 128
run e1_512 timeout 300 python bench.py $B --batch 512
run e1_4096 timeout 300 python bench.py $B --batch 4096
run e8_1024 timeout 300 python bench.py $B --batch 1024 --spec-eps 8
run e8_128 timeout 300 python bench.py $B --batch 128 --spec-eps 8
run e1_1024_rounds RATILQR_FUSED=0 timeout 300 python bench.py $B --batch 1024
