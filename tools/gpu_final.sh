#!/bin/bash
# full GPU suite + smoke + the driver's default bench command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log; t1=$(date +%s); echo "pytest wall $((t1-t0)) s"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
t2=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
t3=$(date +%s); echo "bench wall $((t3-t2)) s"
python3 - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"])
s8=d["secondary_spec_eps8"]; print("e8", round(s8["value"]), s8["values_identical_to_primary"], s8.get("roofline"))
print("shard", {k:round(v["ms_per_batch"],4) for k,v in d["shard_latency_ms"].items()}); print("nm", d["secondary_nm"])
PY
