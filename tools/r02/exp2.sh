#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp2; mkdir -p $O
cd $R
python tools/spmm_bench.py --shapes pubmed500a,pubmed500b,pubmed500c --rounds 5 \
  --variants v2:0:2:16:pet,v2:0:2:8:pet,v2:0:1:0:p,v2:0:2:16:pt > $O/bench_ld.txt 2>&1
python tools/spmm_bench.py --shapes preg4_500a,preg8_500a --rounds 5 \
  --variants v2:0:2:16:pet,v2:0:2:8:pet > $O/bench_preg.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
