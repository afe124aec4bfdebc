#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_bench1; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 50 --warmup 5 > $O/pubmed.json 2> $O/pubmed.err
timeout 600 python bench.py --workload zinc --steps 20 --warmup 3 --no-extra > $O/zinc.json 2> $O/zinc.err
timeout 300 python bench.py --workload zinc --batch-graphs 128 --steps 200 --warmup 20 --no-extra > $O/zinc128.json 2> $O/zinc128.err
timeout 300 python bench.py --workload cora --steps 50 --warmup 5 --no-extra > $O/cora.json 2> $O/cora.err
