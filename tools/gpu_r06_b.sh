#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 300 python tools/duo_time.py 1 16 128 > $O/duo_time.txt 2>&1; cat $O/duo_time.txt
timeout 900 python -m pytest tests/test_gpu_psweep.py -m gpu -x -q > $O/pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_b.log
tail -15 $O/pytest_b.log
