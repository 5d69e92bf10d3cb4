#!/bin/bash
# A/B settings on the GPU box: prints Mpix/s and per-kernel us/view for each environment setting given as an argument.
run() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['per_kernel_us_per_view']
print('%-40s %7.1f Mpix/s  busy/step %.0f/%.0f us  ' % ('$*', d['value'], r['gsr_kernels_busy_us_per_step'], r['step_us']) + ' '.join('%s=%.1f' % (a.replace('preprocess','pre').replace('render_','').replace('radix_','').replace('tile_','').replace('_entries',''), b) for a, b in k.items()))"; }
for cfg in "$@"; do run $cfg; done
