#!/bin/bash
# E = 8 on one GPU: parity tests of the speculative paths, then the bench line per variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest_e8.log 2>&1; echo "pytest rc=$?" >> $O/pytest_e8.log
tail -4 $O/pytest_e8.log
B="--steps 10 --warmup 3 --no-cpu --no-second"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B --batch 1024 --spec-eps 8 > $O/bench_e8_$name.json 2> $O/bench_e8_$name.err
python3 - <<PY
import json
d=json.loads(open("$O/bench_e8_$name.json").read().strip().splitlines()[-1])
print("$name", round(d["value"]), round(d["ms_per_step"],4), {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items() if v}, d["config"].get("mean_iters"))
PY
}
run multi X=1
run stage RATILQR_FLY_MULTI=0
run nofly RATILQR_FLY=0
