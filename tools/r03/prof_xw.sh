#!/bin/bash
# rocprofv3 kernel stats of the layer-1 micro-benchmark (tools/r03/xw_bench.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_xw
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o xw -- python $GRAFT_REPO_ROOT/tools/r03/xw_bench.py ${1:-pubmed} > $OUT/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True) or glob.glob("$OUT/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
