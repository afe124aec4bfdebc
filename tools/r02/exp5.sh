#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp5; mkdir -p $O
cd $R
timeout 600 python tools/spmm_bench.py --shapes pdiag_zero,pdiag_self,pdiag_near,pdiag_col2k,pubmed500a --rounds 5 \
  --variants v2:1:1:16:pet,v2:2:1:16:pet,v2:1:1:8:pet,v2:0:1:16:pEt,v2:1:1:0:pe,v2:0:1:0:p > $O/bench_diag.txt 2>&1
