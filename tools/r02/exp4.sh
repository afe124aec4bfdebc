#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp4; mkdir -p $O
cd $R
timeout 300 python tools/spmm_bench.py --shapes pubmed500a --rounds 7 \
  --variants v2:0:1:16:pEt,v2:0:1:16:pet,v2:1:1:16:pet,v2:2:1:16:pet,v2:0:1:8:pet,v2:1:1:8:pet,v2:2:1:8:pet,v2:2:1:8:pew8t,v2:1:1:8:pew8t,v2:2:1:16:pew8t,v2:2:2:8:pew8t,v2:2:0:8:pew8t > $O/bench_pubmed.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes pubmed32,cora1433a,citeseer3703a --rounds 5 \
  --variants v2:0:1:0:pEt,v2:0:1:0:pet,v2:1:1:0:pet,v2:2:1:0:pet,v2:1:1:0:pew8t > $O/bench_cc.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes zincb39,zincb32 --rounds 5 \
  --variants v2:0:1:0:p,v2:0:1:0:pew4,v2:1:1:0:pew4,v2:2:1:0:pew4,v2:1:1:0:pew8 > $O/bench_zinc.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes preg4_500a,preg8_500a --rounds 3 \
  --variants v2:0:1:16:pet,v2:0:1:8:pet > $O/bench_preg.txt 2>&1
