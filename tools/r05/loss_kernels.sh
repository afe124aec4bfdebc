#!/bin/bash
# per-kernel times of the fused loss inside the captured steps (rocprofv3), Pubmed and ZINC batch 4096
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p $R/gpurun_out/r05x
for w in pubmed zinc; do
  tools/prof_bench.sh r05x/prof_$w --workload $w --steps 30 --warmup 3 | grep -E "bce_|total kernel"
done
