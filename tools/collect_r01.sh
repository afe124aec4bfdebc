#!/bin/bash
# Runs on the GPU box: every measurement quoted in DESIGN.md / profiles/README.md -> gpurun_out/r01/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r01
mkdir -p $O
cd $R
python bench.py > $O/bench_pubmed.json 2> $O/bench_pubmed.err
for w in cora citeseer zinc; do python bench.py --workload $w --no-extra 2>/dev/null | tail -1 > $O/bench_$w.json; done
python bench.py --workload rmat --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rmat_s24_1gpu.json
for w in pubmed cora zinc; do
  tools/prof_bench.sh r01/prof_$w --workload $w --steps 30 --warmup 3 > $O/${w}_step_kernel_stats_top.txt
done
export PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
export PMC_FILTER=spmm
for sh in pubmed500 pubmed32 cora1433 citeseer3703 zincb39 zinc32 zinc39; do
  tools/pmc.sh r01/pmc_$sh tools/spmm_one.py --shape $sh --iters 5 > $O/pmc_$sh.txt
done
tools/pmc.sh r01/pmc_pubmed500_plain tools/spmm_one.py --shape pubmed500 --iters 5 --plain > $O/pmc_pubmed500_plain.txt
python tools/linear_bench.py --rows 256 2>/dev/null > $O/linear_bench.txt
python tools/spmm_bench.py --shapes pubmed500a,pband500 --variants v2:0:1:0,v2:0:1:0:t,v2:0:1:0:et --rounds 5 2>/dev/null > $O/spmm_bench_pubmed.txt
ls -la $O
