#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pets.py tests/test_gpu_ileqg.py tests/test_gpu_isposdef_band.py tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/pets_bench.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/kt_pets -o runc --output-format csv -- python3 tools/pets_bench.py > /dev/null 2>&1
f=$(find $O/prof/kt_pets -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -4
