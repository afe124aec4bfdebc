#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp7; mkdir -p $O
cd $R
timeout 300 python tools/r02/write_floor.py > $O/write_floor.txt 2>&1
