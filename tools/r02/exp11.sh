#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp11; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmm" > $O/pytest_spmm.txt 2>&1
for bpc in 8 4 2; do
timeout 300 python tools/spmm_bench.py --knobs spmm_ell_bpc=$bpc --shapes pdiag_zero,pdiag_col2k,pubmed500a,pubmed500b --rounds 5 \
  --variants v2:1:1:16:pet,v2:2:1:16:pet,v2:1:1:8:pet,v2:1:1:8:pew8t,v2:1:2:16:pet,v2:1:0:16:pet > $O/bench_pubmed_bpc$bpc.txt 2>&1
done
timeout 300 python tools/spmm_bench.py --shapes pubmed32,cora1433a,citeseer3703a --rounds 5 \
  --variants v2:0:1:0:pEt,v2:1:1:0:pet,v2:2:1:0:pet > $O/bench_cc.txt 2>&1
timeout 300 python tools/spmm_bench.py --shapes zincb39,zincb32 --rounds 5 \
  --variants v2:0:1:0:p,v2:1:1:0:pew4,v2:2:1:0:pew4 > $O/bench_zinc.txt 2>&1
timeout 300 python tools/r02/overlap.py > $O/overlap.txt 2>&1
