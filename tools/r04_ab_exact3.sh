#!/bin/bash
# exact per-quad lists, third pass: the tracking walk as two halves of 64 (clean loop), against the same tree without exact lists
O=gpurun_out/r04n; mkdir -p $O; rm -f $O/ab3.txt
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or random_scenes or randomised or early_termination or full_size or fused_pair or per_view or bench_step or one_call" 2>&1 | tail -3 ) > $O/pytest3.log 2>&1; cat $O/pytest3.log
for V in 1 8; do
  echo "== views $V" >> $O/ab3.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_noexact.so libgsr_hip.so >> $O/ab3.txt 2>&1
done
echo "== frozen colours, 8 views" >> $O/ab3.txt
bash tools/ab_libs.sh "--views 8 --no-optimizer --frozen-colours" libgsr_noexact.so libgsr_hip.so >> $O/ab3.txt 2>&1
cut -c1-220 $O/ab3.txt
