#!/bin/bash
# Runs on the GPU box: the round-5 measurements quoted in DESIGN.md / profiles/README.md -> gpurun_out/r05c/
# (tools/publish_profiles.py r05c files them under profiles/ as r05_*).   usage: tools/r05/collect.sh [part ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
PARTS=${@:-bench prof pmc micro}
for part in $PARTS; do
case $part in
bench)
  timeout 900 python bench.py > $O/bench_pubmed.json 2> $O/bench_pubmed.err
  timeout 600 python bench.py --layer1 reference --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pubmed_reference_order.json
  for w in cora citeseer zinc vgae; do timeout 600 python bench.py --workload $w --no-extra 2>/dev/null | tail -1 > $O/bench_$w.json; done
  # (cora / citeseer: --features auto compresses their constant input features; the same steps on the dense features:)
  for w in cora citeseer; do timeout 600 python bench.py --workload $w --features dense --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${w}_dense_features.json; done
  timeout 600 python bench.py --workload zinc --batch-graphs 128 --steps 300 --warmup 30 --no-extra 2>/dev/null | tail -1 > $O/bench_zinc128.json
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 900 python bench.py --workload rmat --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rmat_s24_1gpu.json
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 timeout 900 python bench.py --workload rmat --steps 5 --warmup 2 --no-cpu-baseline --layer-order aggregate-first 2>/dev/null | tail -1 > $O/bench_rmat_s24_1gpu_aggregate_first.json
  ;;
prof)
  for w in pubmed cora citeseer vgae zinc; do
    timeout 600 tools/prof_bench.sh r05c/prof_$w --workload $w --steps 30 --warmup 3 > $O/${w}_step_kernel_stats_top.txt
  done
  timeout 600 tools/prof_bench.sh r05c/prof_zinc128 --workload zinc --batch-graphs 128 --steps 200 --warmup 20 > $O/zinc128_step_kernel_stats_top.txt
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 900 tools/prof_bench.sh r05c/prof_rmat --workload rmat --steps 5 --warmup 2 --no-cpu-baseline > $O/rmat_step_kernel_stats_top.txt
  ;;
pmc)
  export PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  export PMC_FILTER="xw_fwd xtg spmm"
  for sh in pubmed; do
    tools/pmc.sh r05c/pmc_xwfwd_$sh tools/r03/xw_one.py --shape $sh --op fwd > $O/pmc_xwfwd_$sh.txt
    tools/pmc.sh r05c/pmc_xwgrad_$sh tools/r03/xw_one.py --shape $sh --op wgrad > $O/pmc_xwgrad_$sh.txt
  done
  export PMC_FILTER=spmm
  for sh in pubmed500 pubmed32 zincb39; do
    tools/pmc.sh r05c/pmc_$sh tools/spmm_one.py --shape $sh --iters 5 > $O/pmc_$sh.txt
  done
  PMC_TIMEOUT=400 tools/pmc.sh r05c/pmc_rmat32 tools/spmm_one.py --shape rmat32 --rmat-scale 24 --iters 3 > $O/pmc_rmat32.txt
  PMC_TIMEOUT=400 tools/pmc.sh r05c/pmc_rmat16 tools/spmm_one.py --shape rmat16 --rmat-scale 24 --iters 3 > $O/pmc_rmat16.txt
  ;;
micro)
  timeout 300 python tools/r03/xw_bench.py 2>/dev/null > $O/xw_bench.txt
  timeout 300 python tools/r04/xtg_probe.py 2>/dev/null > $O/xtg_probe.txt
  timeout 300 python tools/r04/spx_bench.py 2>/dev/null > $O/spx_bench.txt
  timeout 300 python tools/r04/tall_bench.py 2>/dev/null > $O/tall_bench.txt
  timeout 300 python tools/r04/zinc_l1.py 2>/dev/null > $O/zinc_l1.txt
  timeout 300 python tools/r04/loss_condition.py --sym 2>/dev/null > $O/loss_condition.txt
  timeout 300 python tools/bce_bench.py --variants "sym=1,sb=3;sym=1,sb=2;sym=1,sb=1;sym=1,sb=0;sym=0,sb=0,pb=0" --rounds 5 2>/dev/null > $O/bce_bench_pubmed.txt
  timeout 300 python tools/bce_bench.py --n 95000 --variants "sym=1,sb=3;sym=1,sb=2;sym=1,sb=1" --rounds 3 2>/dev/null > $O/bce_bench_zinc.txt
  timeout 300 python tools/bce_bench.py --graph cora --variants "sym=0,sb=3;sym=0,sb=1;sym=0,sb=0" --rounds 5 2>/dev/null > $O/bce_bench_cora.txt
  timeout 300 python tools/r04/plan_build_time.py 2>/dev/null > $O/plan_build_time.txt
  timeout 300 python tools/r05/xw_ab.py xw_p3=0,1 2>/dev/null > $O/xw_ab.txt
  timeout 300 python tools/r05/xw_fwd_sweep.py 2>/dev/null > $O/xw_fwd_parts.txt
  timeout 300 python tools/r05/xw_stamps.py 2>/dev/null > $O/xw_stamps.txt
  timeout 600 python tools/r05/rmat_order.py 2>/dev/null > $O/rmat_order.txt
  ;;
esac
done
# gpurun copies back at most 64 MiB: the per-dispatch traces are not needed (the stats tables are)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
du -sh $O
ls $O
