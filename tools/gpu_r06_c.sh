#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
timeout 300 python tools/gpu_phases_duo.py 128 > $O/phases_duo.txt 2>&1; cat $O/phases_duo.txt
