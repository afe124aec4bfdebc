#!/bin/bash
# round-2 experiment 1: store policy x tile width on the Pubmed F=500 SpMM
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_exp1; mkdir -p $O
cd $R
python tools/spmm_bench.py --shapes pubmed500a --rounds 7 \
  --variants v2:0:1:16:pet,v2:0:2:16:pet,v2:0:0:16:pet,v2:0:1:8:pet,v2:0:2:8:pet,v2:0:0:8:pet,v2:0:2:8:pt,v2:0:2:16:pt,v2:1:2:8:pet,v2:2:2:8:pet,v2:0:2:4:pet,v2:0:2:24:pet > $O/bench_pubmed.txt 2>&1
python tools/spmm_bench.py --shapes cora1433a,citeseer3703a --rounds 5 \
  --variants v2:0:1:0:pet,v2:0:2:0:pet > $O/bench_cc.txt 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" bash tools/pmc.sh r02_exp1/pmc_tv8_sc1 tools/spmm_one.py --shape pubmed500 --knobs spmm_nt=2,spmm_tile_vecs=8 > $O/pmc_tv8_sc1.txt 2>&1
PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" bash tools/pmc.sh r02_exp1/pmc_tv16_sc1 tools/spmm_one.py --shape pubmed500 --knobs spmm_nt=2,spmm_tile_vecs=16 > $O/pmc_tv16_sc1.txt 2>&1
