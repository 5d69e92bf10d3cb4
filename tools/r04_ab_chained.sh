#!/bin/bash
# tile-range scan chained into the column-prefix launch (one launch less) vs bin_scan as a launch of its own
O=gpurun_out/r04x; mkdir -p $O; rm -f $O/ab.txt
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for A in "--views 1 --no-optimizer" "--views 4 --no-optimizer" "--views 8"; do
  echo "== $A" >> $O/ab.txt
  bash tools/ab_env.sh "$A" "GSR_BIN_CHAINED_SCAN=0" "GSR_BIN_CHAINED_SCAN=1" >> $O/ab.txt 2>&1
done
echo "== config 5" >> $O/ab.txt
for i in 1 2; do for c in 0 1; do
  GSR_BIN_CHAINED_SCAN=$c python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chained=$c', d['ms_per_step'], d['roofline']['per_kernel_us_per_frame'])" >> $O/ab.txt 2>&1
done; done
cut -c1-230 $O/ab.txt
