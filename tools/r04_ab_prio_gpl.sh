#!/bin/bash
# round 4: (a) rank-based ticket base priority sweep in render_bwd, (b) the Gaussian-per-lane prototype (tools/micro/gpl_bwd)
mkdir -p gpurun_out/r04e; O=gpurun_out/r04e; rm -f $O/ab.txt
./tools/micro/gpl_bwd > $O/gpl_bwd.txt 2>&1; echo "rc=$?" >> $O/gpl_bwd.txt
./tools/micro/gpl_bwd 4460 372 >> $O/gpl_bwd.txt 2>&1; echo "rc=$?" >> $O/gpl_bwd.txt
cat $O/gpl_bwd.txt
for V in 1 2 4; do
  echo "== views $V: render_bwd base priority 1 for the longest k/16 of the busy tickets" >> $O/ab.txt
  bash tools/ab_env.sh "--views $V --no-optimizer" "GSR_BWD_PRIO_FRAC16=0" "GSR_BWD_PRIO_FRAC16=4" "GSR_BWD_PRIO_FRAC16=8" "GSR_BWD_PRIO_FRAC16=10" "GSR_BWD_PRIO_FRAC16=12" >> $O/ab.txt 2>&1
done
echo "== views 1: by length" >> $O/ab.txt
bash tools/ab_env.sh "--views 1 --no-optimizer" "GSR_BWD_PRIO_LEN=0" "GSR_BWD_PRIO_LEN=200" "GSR_BWD_PRIO_LEN=250" "GSR_BWD_PRIO_LEN=300" "GSR_BWD_PRIO_LEN=350" >> $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-150
