#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp9; mkdir -p $O
cd $R
timeout 600 python tools/r02/overlap2.py > $O/overlap2.txt 2>&1
