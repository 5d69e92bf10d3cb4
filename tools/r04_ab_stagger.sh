#!/bin/bash
# round 4: staggered tile walks in bin_count / bin_emit (LDS same-address conflicts) + the one-rank RCCL leg of bench.py
O=gpurun_out/r04j; mkdir -p $O; rm -f $O/ab.txt
( python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "rccl" 2>&1 | tail -5 ) > $O/pytest_rccl.log 2>&1; cat $O/pytest_rccl.log
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "tile_row or goldens or batched or config5 or randomised or radix" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; cat $O/pytest.log
for round in 1 2 3; do
for lib in libgsr_nostag.so libgsr_hip.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg5 ms/frame %.3f input-order %.3f' % (d['ms_per_step'], d['ms_per_step_input_order']), d['roofline']['per_kernel_us_per_frame'])" >> $O/ab.txt
done; done
for V in 1 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_nostag.so libgsr_hip.so >> $O/ab.txt 2>&1
done
cut -c1-235 $O/ab.txt
