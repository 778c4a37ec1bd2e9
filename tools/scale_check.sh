#!/bin/bash
# Multi-GPU readiness check: on whatever the box has, run `bench.py --gpus N` for N = 1, 2, 4, 8 up to the number of
# visible devices and assert, per N:
#   * rccl_ranks == N (a real rank census through the collective);
#   * N = 1 reproduces the single-GPU value within 2 % of a second N = 1 run;
#   * the gathered costs are identical across N (bench.py asserts the gather; here: feasible fraction / mean iterations equal);
# then print MEASURED value(N) beside the shard-latency PROJECTION 1024 / (shardN_ms + gather) from the N = 1 line.  The projection is
# printed for comparison only -- it is never a result.       usage: tools/scale_check.sh [steps]      (on a GPU box; one process per GPU)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
STEPS=${1:-20}
OUT=${R}/gpurun_out/scale_check; mkdir -p "$OUT"
NDEV=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "visible devices: $NDEV"
python3 bench.py --gpus 1 --steps "$STEPS" --warmup 3 --no-cpu > "$OUT/n1.json" 2> "$OUT/n1.err" || { echo "N=1 failed"; tail -5 "$OUT/n1.err"; exit 1; }
for N in 1 2 4 8; do
  [ "$N" -gt "$NDEV" ] && break
  python3 bench.py --gpus "$N" --steps "$STEPS" --warmup 3 --no-cpu --no-second > "$OUT/s$N.json" 2> "$OUT/s$N.err" || { echo "N=$N failed"; tail -5 "$OUT/s$N.err"; exit 1; }
done
python3 - "$OUT" "$NDEV" <<'PY'
import json, sys, os
out, ndev = sys.argv[1], int(sys.argv[2])
last = lambda p: json.loads([l for l in open(p) if l.startswith("{")][-1])
n1 = last(os.path.join(out, "n1.json"))
proj = {1: n1["value"]}
for N, key in ((2, "shard512_ms"), (4, "shard256_ms"), (8, "shard128_ms")):
    if key in n1:
        proj[N] = 1024.0 / ((n1[key] + 0.025) * 1e-3)          # + ~25 us all-gather (assumed; unmeasured until an N > 1 node exists)
ok = True
ref = None
print(f"{'N':>2} {'measured solves/s':>18} {'projected (not a result)':>26} {'rccl_ranks':>10} feasible mean_iters")
for N in (1, 2, 4, 8):
    p = os.path.join(out, f"s{N}.json")
    if N > ndev or not os.path.exists(p):
        print(f"{N:>2} {'-- no device --':>18} {proj.get(N, float('nan')):>26.0f}")
        continue
    r = last(p)
    stats = (r["config"]["feasible_fraction"], r["config"]["mean_iters"], r["config"]["mean_ls_evals"])
    ref = ref or stats
    good = r["rccl_ranks"] == N and r["n_gpus"] == N and stats == ref
    if N == 1:
        good = good and abs(r["value"] / n1["value"] - 1.0) < 0.02
    ok = ok and good
    print(f"{N:>2} {r['value']:>18.0f} {proj.get(N, float('nan')):>26.0f} {r['rccl_ranks']:>10} {stats[0]:.4f} {stats[1]:.3f} {'ok' if good else 'MISMATCH'}")
print("scale_check:", "PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
PY
