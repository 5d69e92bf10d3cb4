#!/bin/bash
# exact per-quad lists, final form: full GPU suite, then the A/B against the same tree built without them (GSR_EXACT_LISTS=0, 96-entry batches at 5/CU)
O=gpurun_out/r04p; mkdir -p $O; rm -f $O/ab.txt
export GSR_ROW_MARGINS_LOG=$PWD/$O/row_margins.log
( time python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
for A in "--views 1 --no-optimizer" "--views 4 --no-optimizer" "--views 8 --no-optimizer" "--views 8 --no-optimizer --frozen-colours" "--views 8"; do
  echo "== $A" >> $O/ab.txt
  bash tools/ab_libs.sh "$A" libgsr_noexact.so libgsr_hip.so >> $O/ab.txt 2>&1
done
echo "== config 5 (forward only: no tracking)" >> $O/ab.txt
for i in 1 2; do for lib in libgsr_noexact.so libgsr_hip.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['metric'], d['value'], d.get('ms_per_step'))" >> $O/ab.txt 2>&1
done; done
cut -c1-220 $O/ab.txt
