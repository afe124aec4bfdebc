#!/bin/bash
# Runs on the GPU box: every measurement quoted in DESIGN.md / profiles/README.md -> gpurun_out/<tag>/
# usage: tools/collect.sh <tag>        (tag = r02, ...; tools/publish_profiles.py <tag> files the results)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_pubmed.json 2> $O/bench_pubmed.err
for w in cora citeseer zinc; do timeout 600 python bench.py --workload $w --no-extra 2>/dev/null | tail -1 > $O/bench_$w.json; done
# BASELINE config 5: VGAE on Citeseer, bf16 feature storage
timeout 600 python bench.py --workload vgae --no-extra 2>/dev/null | tail -1 > $O/bench_vgae.json
timeout 600 python bench.py --workload zinc --batch-graphs 128 --steps 300 --warmup 30 --no-extra 2>/dev/null | tail -1 > $O/bench_zinc128.json
# the same inductive steps launched eagerly (no captured step): what the HIP graph buys
timeout 600 python bench.py --workload zinc --no-hipgraph --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_zinc_eager.json
timeout 600 python bench.py --workload zinc --batch-graphs 128 --steps 300 --warmup 30 --no-hipgraph --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_zinc128_eager.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 900 python bench.py --workload rmat --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rmat_s24_1gpu.json
for w in pubmed cora zinc; do
  timeout 600 tools/prof_bench.sh $TAG/prof_$w --workload $w --steps 30 --warmup 3 > $O/${w}_step_kernel_stats_top.txt
done
timeout 600 tools/prof_bench.sh $TAG/prof_zinc128 --workload zinc --batch-graphs 128 --steps 200 --warmup 20 > $O/zinc128_step_kernel_stats_top.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 900 tools/prof_bench.sh $TAG/prof_rmat --workload rmat --steps 5 --warmup 2 --no-cpu-baseline > $O/rmat_step_kernel_stats_top.txt
export PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
export PMC_FILTER=spmm
for sh in pubmed500 pubmed32 cora1433 citeseer3703 zincb39 zinc32 zinc39; do
  tools/pmc.sh $TAG/pmc_$sh tools/spmm_one.py --shape $sh --iters 5 > $O/pmc_$sh.txt
done
PMC_TIMEOUT=300 tools/pmc.sh $TAG/pmc_rmat32 tools/spmm_one.py --shape rmat32 --rmat-scale 24 --iters 3 > $O/pmc_rmat32.txt
tools/pmc.sh $TAG/pmc_pubmed500_plain tools/spmm_one.py --shape pubmed500 --iters 5 --plain > $O/pmc_pubmed500_plain.txt
timeout 300 python tools/linear_bench.py --rows 256 2>/dev/null > $O/linear_bench.txt
timeout 300 python tools/spmm_bench.py --shapes pubmed500a,pband500 --variants v2:0:1:0:p,v2:0:1:0:pt,v2:0:1:0:pEt,v2:1:1:0:pet --rounds 5 2>/dev/null > $O/spmm_bench_pubmed.txt
timeout 300 python tools/spmm_bench.py --shapes pdiag_zero,pdiag_col2k --variants v2:0:1:0:p,v2:1:1:16:pet --rounds 5 2>/dev/null > $O/spmm_bench_diag.txt
timeout 300 python tools/bce_bench.py --variants "sym=1;sym=0;sym=0,sb=0,pb=0" --rounds 5 2>/dev/null > $O/bce_bench_pubmed.txt
timeout 300 python tools/bce_bench.py --n 94752 --variants "sym=1;sym=1,sri=2;sym=0" --rounds 3 2>/dev/null > $O/bce_bench_zinc.txt
timeout 200 tools/probes/bin/gather_l2 > $O/probe_gather_l2.txt 2>&1
timeout 200 tools/probes/bin/gather_l2b > $O/probe_gather_l2b.txt 2>&1
timeout 100 tools/probes/bin/valu_rate > $O/probe_valu_rate.txt 2>&1
timeout 100 tools/probes/bin/inst_cost > $O/probe_inst_cost.txt 2>&1
timeout 100 tools/probes/bin/mfma32_check > $O/probe_mfma32_check.txt 2>&1
ls -la $O
