#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp6; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmm" > $O/pytest_spmm.txt 2>&1
timeout 600 python tools/spmm_bench.py --shapes pdiag_zero,pubmed500a --rounds 7 \
  --variants v2:0:1:16:pEt,v2:1:1:16:pet,v2:2:1:16:pet,v2:1:1:16:pest,v2:2:1:16:pest,v2:1:1:8:pest,v2:2:1:8:pest,v2:1:1:8:pesw8t,v2:2:1:8:pesw8t,v2:2:1:16:pesw8t > $O/bench_pubmed.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes pubmed32,cora1433a,citeseer3703a --rounds 5 \
  --variants v2:0:1:0:pEt,v2:1:1:0:pet,v2:1:1:0:pest,v2:2:1:0:pest > $O/bench_cc.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes zincb39,zincb32 --rounds 5 \
  --variants v2:0:1:0:p,v2:1:1:0:pew4,v2:1:1:0:pesw4,v2:2:1:0:pesw4 > $O/bench_zinc.txt 2>&1
