#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
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
for b in 128 256 512; do
run e8_${b}_auto timeout 300 python bench.py $B --batch $b --spec-eps 8
run e8_${b}_rounds RATILQR_BLOCK=0 timeout 300 python bench.py $B --batch $b --spec-eps 8
done
run e4_256_auto timeout 300 python bench.py $B --batch 256 --spec-eps 4
run e4_256_rounds RATILQR_BLOCK=0 timeout 300 python bench.py $B --batch 256 --spec-eps 4
run e2_512_auto timeout 300 python bench.py $B --batch 512 --spec-eps 2
run e2_512_rounds RATILQR_BLOCK=0 timeout 300 python bench.py $B --batch 512 --spec-eps 2
