#!/bin/bash
# PMC passes over tools/profile_eval.py (counters only: no trace domains besides kernel-trace)
# usage: tools/pmc_run.sh <tag> [c2|c4|c5]   -> gpurun_out/<tag>/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$1_*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$1_$i -- python $R/tools/profile_eval.py 8 0 ${2:-c2} > $OUT/pass$i.log 2>&1
done
python $R/tools/pmc_summary.py /tmp pmc_$1_ > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
