#!/bin/bash
# A/B blend-kernel scheduling on the GPU box: prints Mpix/s and per-kernel us/view for each setting.
run() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_us_per_view']
print('%-50s %7.1f Mpix/s  fwd %6.1f  bwd %6.1f  sort %5.1f ranges+order %4.1f  all %s' % ('$*', d['value'], k['render_fwd'], k['render_bwd'], k['tile_sort'], k.get('tile_ranges',0), ' '.join('%s=%.1f'%(a,b) for a,b in k.items())))"; }
for cfg in "$@"; do run $cfg; done
