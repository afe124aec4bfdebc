#!/bin/bash
# SQ counters of the tall dense passes (csrc/tall.hip) on 2^24 rows (tools/r04/tall_bench.py): is linear2_rows_kernel /
# gcn2_bwd_rows_kernel waiting for memory or for the fp32 matrix pipe?   -> gpurun_out/r06/tall_sq/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06/tall_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/p1 -o pmc -- python $R/tools/r04/tall_bench.py > $O/p1.log 2>&1
timeout -k 5 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/p2 -o pmc -- python $R/tools/r04/tall_bench.py > $O/p2.log 2>&1
python - <<PY > $O/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "linear2_rows" not in k and "gcn2_bwd_rows" not in k: continue
        agg[k[:60] + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("tall dense passes, 2^24 rows (tools/r04/tall_bench.py: all rows | first 5.03 M rows | list mode with 30 % live rows); per launch means")
print("units: GRBM_GUI_ACTIVE summed over 8 XCDs; a v_mfma_f32_32x32x2_f32 holds the matrix pipe 64 cycles (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA below)")
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if not cyc: continue
    mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print(f"    {cyc:10.0f} cycles per SIMD ({cyc / 2.4e3:7.1f} us); MFMAs per SIMD {m.get('SQ_INSTS_MFMA', 0) / 1024:9.0f}, busy cycles per MFMA "
          f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):5.1f}; matrix pipe busy {100 * mf / cyc:5.1f} %")
    print(f"    waves' resident time: issuing {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.1f} %, stalled at issue {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:4.1f} %, "
          f"parked at s_waitcnt {100 * m.get('SQ_WAIT_ANY', 0) / wc:4.1f} %; waves {m.get('SQ_WAVES', 0):.0f}")
PY
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
