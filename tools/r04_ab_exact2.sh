#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O; rm -f $O/ab2.txt
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "goldens or random_scenes or randomised or early_termination or full_size or fused_pair or per_view or bench_step or one_call" 2>&1 | tail -3 ) > $O/pytest2.log 2>&1; cat $O/pytest2.log
for V in 1 8; do
  echo "== views $V" >> $O/ab2.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_r04base.so libgsr_hip.so libgsr_un6.so libgsr_un8.so >> $O/ab2.txt 2>&1
done
cut -c1-220 $O/ab2.txt
