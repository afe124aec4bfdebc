#!/bin/bash
# usage: tools/prof_tool.sh <name> <script under the repo> [args...]  -> gpurun_out/<name>/ kernel stats of that script
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; SCRIPT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$NAME -o run -- python $R/$SCRIPT "$@" > $R/gpurun_out/$NAME.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$NAME/run_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time (all calls) %.1f us" % (tot/1e3))
for r in rows[:14]:
    nm=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%-86s calls %5s avg %9.1f us  %5.1f%%" % (nm[:86], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
