#!/bin/bash
# the driver's default bench command, timed, + the distributed bench tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
s=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
e=$(date +%s.%N); echo "bench wall $(echo "$e - $s" | bc) s"
python3 - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"])
print("steady", d["steady_state"]["wall_s"], d["steady_state"]["value_from_median"])
s8=d["secondary_spec_eps8"]; print("e8", round(s8["value"]), s8["values_identical_to_primary"], s8.get("kernel_ms_per_step"), s8.get("roofline"))
print("nonlin", d["secondary_nonlinear"]["value"]); print("large", {k:round(v["value"]) for k,v in d["secondary_large_batch"].items()})
print("shard", d["shard_latency_ms"]); print("pets", {k:(round(v["trajectories_per_s"]), round(v["kernel_ms"],4), v["roofline"]["frac"]) for k,v in d["secondary_pets"]["runs"].items()})
print("nm", d["secondary_nm"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]); print("rounds", d["round_based_path"]["kernel_ms_per_step"])
print("config", d["config"])
PY
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_pets.py tests/test_gpu_nm.py -m gpu -x -q 2>&1 | tail -4
