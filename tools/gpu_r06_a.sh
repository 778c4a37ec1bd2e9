#!/bin/bash
# round 6, first call: the new parity tests + the cross-workgroup hand-off microbenchmark
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_psweep.py tests/test_gpu_multi.py "tests/test_gpu_pets.py" -m gpu -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" >> $O/pytest_a.log
tail -15 $O/pytest_a.log
timeout 120 tools/ubench/xwg_handoff > $O/xwg_handoff.txt 2>&1; cat $O/xwg_handoff.txt
