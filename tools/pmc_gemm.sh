#!/bin/bash
# PMC passes (separate, no sys-trace) over one GEMM shape: tools/pmc_gemm.sh M N K hint out.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pm$i -o p --output-format csv -- python $R/tools/bench_gemm_one.py $1 $2 $3 $4 > $R/gpurun_out/pm$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmcstat.py /tmp/pm*/p_counter_collection.csv --match gemm_kernel > $R/gpurun_out/$5
cat $R/gpurun_out/$5
