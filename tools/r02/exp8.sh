#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp8; mkdir -p $O
cd $R
timeout 600 python tools/r02/overlap.py > $O/overlap.txt 2>&1
