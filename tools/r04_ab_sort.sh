#!/bin/bash
# round 4: the wave bitonic sort with the element layout e = lane * NR + r (fewer cross-lane steps); tests + A/B against the round-start library
O=gpurun_out/r04h; mkdir -p $O; rm -f $O/ab.txt
( python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "selftest or tile_sort or huge_tile or tile_row or goldens or randomised or radix or config5 or early_termination" 2>&1 | tail -3 ) > $O/pytest.log 2>&1; cat $O/pytest.log
for V in 4 8; do
  echo "== views $V" >> $O/ab.txt
  bash tools/ab_libs.sh "--views $V --no-optimizer" libgsr_r04base.so libgsr_hip.so >> $O/ab.txt 2>&1
done
for lib in libgsr_r04base.so libgsr_hip.so libgsr_r04base.so libgsr_hip.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg5 ms/frame %.3f' % d['ms_per_step'], d['roofline']['per_kernel_us_per_frame'])" >> $O/ab.txt
done
cut -c1-220 $O/ab.txt
