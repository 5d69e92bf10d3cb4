#!/bin/bash
# SQ counters of ANY kernel(s) of a bench run: tools/prof_kernel_sq.sh "<kernel regex>" "<bench args>"   -> gpurun_out/sq_<tag>.txt
# Two --pmc passes (8 SQ counters each), kernel trace only (no sys/hip/hsa trace domains next to counters).
REGEX=$1; ARGS=$2; TAG=${3:-kernel}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for set in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $O/prof_ksq$i
  rocprofv3 --kernel-trace --output-format csv --pmc $set --kernel-include-regex "$REGEX" -d $O/prof_ksq$i -o run -- \
    python $R/bench.py $ARGS --no-cpu-baseline --no-extras > $O/prof_ksq$i.log 2>&1
done
cd $R
{ python tools/prof_summarize.py pmc $O/prof_ksq1; python tools/prof_summarize.py pmc $O/prof_ksq2; } > $O/sq_$TAG.txt
rm -rf $O/prof_ksq1 $O/prof_ksq2
cat $O/sq_$TAG.txt
