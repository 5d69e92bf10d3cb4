#!/bin/bash
# Round 6: streams NO build of this repository has seen before, on the final tree.  usage: tools/r06_soak_new_seeds.sh "seed:cases:big ..." (default: the round's first fresh set)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_soak_new; mkdir -p $O; rm -f $R/gpurun_out/parity_soak.txt $O/summary.txt
run() { local t0=$(date +%s)
  GSR_SOAK_CASES=$1 GSR_SOAK_SEED=$2 GSR_SOAK_BIG=$3 timeout 3000 python -m pytest $R/tests/test_soak_gpu.py::test_parity_soak -x -q > $O/s$2_b$3.log 2>&1
  echo "cases=$1 seed=$2 big=$3 rc=$? seconds=$(( $(date +%s) - t0 )) :: $(tail -1 $O/s$2_b$3.log)" >> $O/summary.txt; }
for spec in ${1:-11:1500:0 12:1500:0 2718:1500:0 7:200:1}; do IFS=: read s c b <<< "$spec"; run $c $s $b; done
cp $R/gpurun_out/parity_soak.txt $O/parity_soak.txt; cat $O/summary.txt
