#!/bin/bash
# Round 6: cost of the per-Gaussian backward's chain in fp64 (experiment build pb64 = -DPB_REAL=double), one box, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_pb64; mkdir -p $O; : > $O/ab2.txt
for a in "--config 3" "--views 8" "--views 2"; do
  echo "## bench.py $a" >> $O/ab2.txt
  bash $R/tools/ab_libs.sh "$a" $LIBS >> $O/ab2.txt 2>&1
done
cat $O/ab2.txt
