#!/bin/bash
# PMC passes over any of the tools/*.py scripts (GPU box): one counter group per run, each under its own timeout.
# usage: bash tools/pmc_any.sh <tag> <kernel-name-substring> <script.py> [args...]
tag=$1; sub=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_IFETCH SQ_WAVES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/"$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_$i.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out"; exit $rc; fi
  echo "pass $i rc=$rc"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/${tag} $sub > gpurun_out/${tag}_summary.txt 2>&1; cat gpurun_out/${tag}_summary.txt
