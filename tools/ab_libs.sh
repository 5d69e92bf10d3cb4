#!/bin/bash
# A/B of library builds on ONE GPU box (boxes differ by +-2 %): tools/ab_libs.sh "<bench args>" libA.so libB.so ...  (3 alternating rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; ARGS=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    GSR_HIP_LIB=$R/gs-dynamics_amd/csrc/$lib timeout 300 python $R/bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['per_kernel_us_per_launch']
print('%-22s step %.1f us (median %.1f) | ' % ('$lib', r['step_us'], 1e3*d['ms_per_step_event_median']) + ' '.join('%s=%.1f' % (a.replace('preprocess','pre').replace('render_','R').replace('radix_','').replace('tile_','').replace('_entries','').replace('activate','act').replace('_views',''), b) for a, b in k.items()))"
  done
done
