#!/bin/bash
# bench.py on the GPU box into gpurun_out/r04/<name>.json and a one-screen digest:  tools/gpu_bench.sh NAME [bench args] [-- leg ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; mkdir -p gpurun_out/r04
NAME=$1; shift
ARGS=(); LEGS=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; LEGS=("$@"); break; fi; ARGS+=("$1"); shift; done
python bench.py "${ARGS[@]}" > gpurun_out/r04/$NAME.json 2> gpurun_out/r04/$NAME.err || tail -5 gpurun_out/r04/$NAME.err
python tools/bench_brief.py gpurun_out/r04/$NAME.json "${LEGS[@]}"
