#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace: true start / end of every kernel, idle gaps) of one steady-state step with V views.
# usage: tools/view_timeline.sh V [extra bench args]   -> gpurun_out/timeline_vV.txt
V=${1:-1}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_tl
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o run -- python $R/bench.py --views $V --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" > $O/prof_tl.log 2>&1 || true
cd $R
python tools/timeline.py $O/prof_tl 1 > $O/timeline_v$V.txt 2>&1 || true
rm -rf $O/prof_tl
cat $O/timeline_v$V.txt
