#!/bin/bash
# PMC passes over the JVP-only script (GPU box): one counter group per run, each under its own timeout.
# usage: bash tools/pmc_jvp.sh <tag> <GEO_JVP_MID value>
tag=${1:-pmcjvp}; export GEO_JVP_MID=${2:-0}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/exp_jvp_ablate.py prod > $GRAFT_REPO_ROOT/gpurun_out/${tag}_$i.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out"; exit $rc; fi
  echo "pass $i rc=$rc"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/${tag} mid_ > gpurun_out/${tag}_summary.txt 2>&1; cat gpurun_out/${tag}_summary.txt
