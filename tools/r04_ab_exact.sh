#!/bin/bash
# round 4: the backward's per-quad lists from the forward's contribution bytes (GSR_EXACT_LISTS): full GPU suite + A/B against the round-start library
O=gpurun_out/r04k; mkdir -p $O; rm -f $O/ab.txt
( python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; cat $O/pytest.log
for V in 1 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_r04base.so libgsr_hip.so >> $O/ab.txt 2>&1
done
echo "== views 8, frozen colours" >> $O/ab.txt
bash tools/ab_libs.sh "--views 8 --no-optimizer --frozen-colours" libgsr_r04base.so libgsr_hip.so >> $O/ab.txt 2>&1
cut -c1-220 $O/ab.txt
