#!/bin/bash
# rocprofv3 kernel statistics of the fused get_loss step (4 cameras, t > 0), every kernel of the process -- the library's and torch's
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_getloss; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $R/tools/r05_getloss_kernels.py > $O/run.log 2>&1 || true
cd $R
python tools/prof_summarize.py stats $O/prof > $O/kernel_stats.txt 2>&1 || true
head -40 $O/kernel_stats.txt; tail -3 $O/run.log
