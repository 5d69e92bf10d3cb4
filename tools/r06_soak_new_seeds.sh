#!/bin/bash
# Round 6: streams NO build of this repository has seen before (seeds 11, 12, 2718 at 1500 cases; GSR_SOAK_BIG seed 7 at 200), on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_soak_new; mkdir -p $O; rm -f $R/gpurun_out/parity_soak.txt $O/summary.txt
run() { local t0=$(date +%s)
  GSR_SOAK_CASES=$1 GSR_SOAK_SEED=$2 GSR_SOAK_BIG=$3 timeout 3000 python -m pytest $R/tests/test_soak_gpu.py::test_parity_soak -x -q > $O/s$2_b$3.log 2>&1
  echo "cases=$1 seed=$2 big=$3 rc=$? seconds=$(( $(date +%s) - t0 )) :: $(tail -1 $O/s$2_b$3.log)" >> $O/summary.txt; }
run 1500 11 0; run 1500 12 0; run 1500 2718 0; run 200 7 1
cp $R/gpurun_out/parity_soak.txt $O/parity_soak.txt; cat $O/summary.txt
