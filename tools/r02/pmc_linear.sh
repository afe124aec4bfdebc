#!/bin/bash
# counters of the tall-skinny Linear kernels on the Pubmed layer-1 shape (forward, dW)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PMC_FILTER="linear_fwd gemm_stream atb_bf16"
export PMC_SETS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;FETCH_SIZE;GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
PMC_TIMEOUT=150 bash tools/pmc.sh r02_pmc_linear tools/linear_bench.py --rows 0 --only "pubmed L1" > gpurun_out/r02_pmc_linear.txt 2>&1
cat gpurun_out/r02_pmc_linear.txt
