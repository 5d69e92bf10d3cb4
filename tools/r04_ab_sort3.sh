#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O; rm -f $O/ab3.txt
( GSR_NO_TORCH_EXT=1 GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/libgsr_x4dpp.so python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "selftest or tile_sort or huge_tile or tile_row or goldens" 2>&1 | tail -2 ) >> $O/ab3.txt
for round in 1 2; do
for lib in libgsr_hip.so libgsr_x4dpp.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg5 ms/frame %.3f' % d['ms_per_step'], d['roofline']['per_kernel_us_per_frame'])" >> $O/ab3.txt
done; done
bash tools/ab_libs.sh "--views 4 --no-optimizer" libgsr_hip.so libgsr_x4dpp.so >> $O/ab3.txt 2>&1
cut -c1-230 $O/ab3.txt
