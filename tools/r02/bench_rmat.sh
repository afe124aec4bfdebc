#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_bench_rmat; mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
timeout 900 python bench.py --workload rmat --rmat-scale 22 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/rmat22.json 2> $O/rmat22.err
timeout 900 python bench.py --workload rmat --rmat-scale 22 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-overlap > $O/rmat22_noov.json 2> $O/rmat22_noov.err
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload rmat --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/rmat24.json 2> $O/rmat24.err
