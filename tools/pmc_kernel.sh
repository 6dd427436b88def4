#!/bin/bash
# PMC passes over one command: tools/pmc_kernel.sh <kernel substr> <out.txt> -- python ...
sub=$1; out=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pk$i -o p --output-format csv -- "$@" > $R/gpurun_out/pk$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmcstat.py /tmp/pk*/p_counter_collection.csv --match "$sub" > $R/gpurun_out/$out
python $R/tools/kstat.py /tmp/pk1 "$sub" >> $R/gpurun_out/$out
cat $R/gpurun_out/$out
