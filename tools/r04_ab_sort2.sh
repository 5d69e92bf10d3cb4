#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O; rm -f $O/ab.txt
for round in 1 2; do
for lib in libgsr_hip.so libgsr_w5.so libgsr_w6.so; do
  GSR_HIP_LIB=$PWD/gs-dynamics_amd/csrc/$lib python bench.py --config 5 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg5 ms/frame %.3f' % d['ms_per_step'], d['roofline']['per_kernel_us_per_frame'])" >> $O/ab.txt
done; done
cat $O/ab.txt
