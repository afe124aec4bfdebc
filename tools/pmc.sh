#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <python args...>   -- separate --pmc passes (guide: HBM counters alone)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
if [ -z "$PMC_SETS" ]; then
  PMC_SETS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
fi
IFS=';' read -ra SETS <<< "$PMC_SETS"
for CNT in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -o pmc -- python $R/"$1" "${@:2}" > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in "${PMC_FILTER:-spmm bce gemm atb}".split()): continue
        agg[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY
