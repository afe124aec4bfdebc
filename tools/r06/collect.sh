#!/bin/bash
# Runs on the GPU box: the round-6 measurements quoted in DESIGN.md / profiles/README.md -> gpurun_out/r06c/
# (tools/publish_profiles.py r06c files them under profiles/ as r06_*).   usage: tools/r06/collect.sh [part ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
PARTS=${@:-bench prof pmc micro}
for part in $PARTS; do
case $part in
bench)
  timeout 900 python bench.py > $O/bench_pubmed.json 2> $O/bench_pubmed.err
  timeout 600 python bench.py --layer1 reference --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pubmed_reference_order.json
  for w in cora citeseer zinc vgae; do timeout 600 python bench.py --workload $w --no-extra 2>/dev/null | tail -1 > $O/bench_$w.json; done
  timeout 600 python bench.py --workload zinc --batch-graphs 128 --steps 300 --warmup 30 --no-extra 2>/dev/null | tail -1 > $O/bench_zinc128.json
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 900 python bench.py --workload rmat --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rmat_s24_1gpu.json
  ;;
prof)
  for w in pubmed cora citeseer vgae zinc; do
    timeout 600 tools/prof_bench.sh r06c/prof_$w --workload $w --steps 30 --warmup 3 > $O/${w}_step_kernel_stats_top.txt
  done
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 900 tools/prof_bench.sh r06c/prof_rmat --workload rmat --steps 5 --warmup 2 --no-cpu-baseline > $O/rmat_step_kernel_stats_top.txt
  ;;
pmc)
  export PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  export PMC_FILTER="xw_fwd xtg spmm"
  tools/pmc.sh r06c/pmc_xwfwd_pubmed tools/r03/xw_one.py --shape pubmed --op fwd > $O/pmc_xwfwd_pubmed.txt
  tools/pmc.sh r06c/pmc_xwgrad_pubmed tools/r03/xw_one.py --shape pubmed --op wgrad > $O/pmc_xwgrad_pubmed.txt
  export PMC_FILTER=spmm
  for sh in pubmed500 pubmed32 zincb39; do
    tools/pmc.sh r06c/pmc_$sh tools/spmm_one.py --shape $sh --iters 5 > $O/pmc_$sh.txt
  done
  # the Pubmed F = 500 aggregation under other XCD feature-tile widths (TCC_HIT / MISS and FETCH_SIZE per variant)
  for tv in 8 24 -1; do
    tools/pmc.sh r06c/pmc_pubmed500_tv$tv tools/spmm_one.py --shape pubmed500 --iters 5 --knobs spmm_tile_vecs=$tv > $O/pmc_pubmed500_tv$tv.txt
  done
  PMC_TIMEOUT=400 tools/pmc.sh r06c/pmc_rmat32 tools/spmm_one.py --shape rmat32 --rmat-scale 24 --iters 3 > $O/pmc_rmat32.txt
  ;;
losssq)
  tools/r06/loss_sq.sh > /dev/null 2>&1
  cp $R/gpurun_out/r06/loss_sq/summary.txt $O/loss_sq.txt
  cp $R/gpurun_out/r06/loss_sq/wave_spec_overlap.txt $O/probe_wave_spec_overlap.txt
  ;;
micro)
  timeout 300 python tools/bce_bench.py --variants "sri=2,bal=0;sri=2,bal=1;sri=4,bal=0;sri=4,bal=1" --rounds 5 2>/dev/null > $O/bce_bench_pubmed.txt
  for n in 3327 5000 8000 26000 40000; do echo "== N = $n" >> $O/bce_bench_sizes.txt; timeout 300 python tools/bce_bench.py --n $n --variants "sym=1,bal=0,sri=2;sym=2,sri=2,bal=0;sym=2,sri=2,bal=1;sym=2,sri=4,bal=0;sym=2,sri=4,bal=1" --rounds 4 2>/dev/null | grep median >> $O/bce_bench_sizes.txt; done
  timeout 300 python tools/bce_bench.py --n 95000 --variants "sri=2,bal=0;sri=2,bal=2;sri=4,bal=0;sri=4,bal=2" --rounds 3 2>/dev/null > $O/bce_bench_zinc.txt
  timeout 600 python tools/r06/spmm_tile_sweep.py --shape pubmed 2>/dev/null > $O/spmm_tile_sweep_pubmed.txt
  timeout 600 python tools/r06/spmm_tile_sweep.py --shape cora --tiles 0,8,16,32,64,-1 --stores -1 2>/dev/null > $O/spmm_tile_sweep_cora.txt
  [ -s $O/rmat_windows.txt ] || timeout 900 python tools/r06/rmat_windows.py 2>/dev/null > $O/rmat_windows.txt
  ;;
esac
done
# gpurun copies back at most 64 MiB: the per-dispatch traces are not needed (the stats tables are)
find $O $R/gpurun_out/r06 -name "*kernel_trace.csv" -delete 2>/dev/null
find $O $R/gpurun_out/r06 -name "*.db" -delete 2>/dev/null
find $R/gpurun_out/r06 -name "*counter_collection.csv" -size +4M -delete 2>/dev/null
du -sh $O
