#!/bin/bash
# SQ counters of the image-loss kernels inside the fused get_loss step (tools/r05_getloss_kernels.py); two --pmc passes, kernel trace only
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for set in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $O/prof_lsq$i
  timeout 250 rocprofv3 --kernel-trace --output-format csv --pmc $set --kernel-include-regex "image_loss" -d $O/prof_lsq$i -o run -- \
    python $R/tools/r05_getloss_kernels.py > $O/prof_lsq$i.log 2>&1
done
cd $R
{ python tools/prof_summarize.py pmc $O/prof_lsq1; python tools/prof_summarize.py pmc $O/prof_lsq2; } > $O/sq_loss.txt
rm -rf $O/prof_lsq1 $O/prof_lsq2
cat $O/sq_loss.txt
