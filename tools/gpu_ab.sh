#!/bin/bash
# parity-critical tests on the tree's library, then the A/B of build/ab/*.so (tools/ab_bench.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_parity.py tests/test_gpu_ce.py tests/test_gpu_ileqg.py -m gpu -x -q > $O/pytest_ab.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ab.log
tail -5 $O/pytest_ab.log
bash tools/ab_bench.sh "$@"
