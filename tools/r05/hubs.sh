#!/bin/bash
# Citation steps on graphs with the REAL graphs' degree profile (hubs of 100 - 170 neighbours), by plan policy:
# GAE_TABLE_MAXDEG=64 (the rule of rounds 2-4: rows beyond 8 edges go through the segment kernels once a row is longer
# than 64) against the default 1024 (table-only plan: long rows continue from the CSR arrays, the whole wave on one row
# -- spmm_ell.hip ell_long_row, decoder_bce.hip bce_edges_kernel).
mkdir -p gpurun_out/r05_hubs
for w in pubmed cora citeseer; do
  for deg in uniform planetoid; do
    for lim in 64 1024; do
      # (the old rule leaves a hubbed graph without the table-only plans transform-first / sparse features need: the
      #  scripts then run the reference's layer order on dense features -- what `--layer1 reference` times)
      extra=""; if [ $lim = 64 ] && [ $deg = planetoid ]; then extra="--layer1 reference --features dense"; fi
      GAE_TABLE_MAXDEG=$lim python bench.py $extra --workload $w --degrees $deg --steps 300 --warmup 30 --no-cpu-baseline \
        > gpurun_out/r05_hubs/${w}_${deg}_${lim}.json 2> gpurun_out/r05_hubs/${w}_${deg}_${lim}.err
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_hubs/${w}_${deg}_${lim}.json").read().strip().splitlines()[-1])
    print("$w $deg limit $lim: ms_per_step", d["ms_per_step"], "longest_row", d["config"].get("longest_row"), d["config"].get("layer1", "")[:60])
except Exception as e:
    print("$w $deg $lim FAILED", e)
PY
    done
  done
done
