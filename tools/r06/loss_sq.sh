#!/bin/bash
# SQ counters of the fused loss's dense kernel (bce_dense_sym_kernel), Pubmed (N = 19 717) and the
# ZINC-4096 batch size (N = 95 000), plus the calibration kernels of tools/probes/wave_spec_overlap.hip (pure-MFMA, pure-VALU,
# mixed, wave-specialised) under the same counters.  Separate passes of <= 8 SQ counters; no tracing flags beside --pmc.
#   usage (GPU box): tools/r06/loss_sq.sh   -> gpurun_out/r06/loss_sq/{summary.txt,...}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06/loss_sq; mkdir -p $O
mkdir -p tools/probes/bin
hipcc --offload-arch=gfx950 -O3 tools/probes/wave_spec_overlap.hip -o tools/probes/bin/wave_spec_overlap 2>/dev/null
tools/probes/bin/wave_spec_overlap > $O/wave_spec_overlap.txt 2>&1
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES SQ_WAVES"
      "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16")
i=0
for CNT in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $CNT --output-format csv -d $O/pubmed_p$i -o pmc -- python $R/tools/bce_bench.py --graph pubmed --variants "sym=1,sb=3" --rounds 1 > $O/pubmed_p$i.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc $CNT --output-format csv -d $O/zinc_p$i -o pmc -- python $R/tools/bce_bench.py --n 95000 --variants "sym=1,sb=3" --rounds 1 > $O/zinc_p$i.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc $CNT --output-format csv -d $O/probe_p$i -o pmc -- $R/tools/probes/bin/wave_spec_overlap > $O/probe_p$i.log 2>&1
done
python $R/tools/r06/loss_sq_summary.py $O > $O/summary.txt
# the raw per-dispatch tables are large: keep the summary and the logs only
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
