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
run e8_1024 timeout 300 python bench.py $B --batch 1024 --spec-eps 8
run e8_512 timeout 300 python bench.py $B --batch 512 --spec-eps 8
STRESS_S=100 timeout 900 python tools/stress_paths.py 2>&1 | tail -2
timeout 600 python bench.py --no-cpu --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary_spec_eps8']; print('default bench e8', round(s['value']), s['values_identical_to_primary'], 'headline', round(d['value']))"
