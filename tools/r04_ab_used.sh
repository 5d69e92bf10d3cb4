#!/bin/bash
# per-Gaussian used flags: the per-Gaussian backward skips the records of Gaussians no pixel of the view blended
O=gpurun_out/r04w; mkdir -p $O; rm -f $O/ab.txt
export GSR_ROW_MARGINS_LOG=$PWD/$O/row_margins.log
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for A in "--views 1 --no-optimizer" "--views 4 --no-optimizer" "--views 8" "--views 8 --frozen-colours"; do
  echo "== $A" >> $O/ab.txt
  bash tools/ab_env.sh "$A" "GSR_NO_USED_FLAGS=1" "GSR_NO_USED_FLAGS=0" >> $O/ab.txt 2>&1
done
cut -c1-230 $O/ab.txt
