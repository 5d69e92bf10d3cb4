#!/bin/bash
# A/B of environment settings on ONE GPU box: tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (3 alternating rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; ARGS=$1; shift
for round in 1 2 3; do
  for cfg in "$@"; do
    env $cfg timeout 300 python $R/bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['per_kernel_us_per_launch']
print('%-34s step %.1f us (median %.1f) | ' % ('$cfg', r['step_us'], 1e3*d['ms_per_step_event_median']) + ' '.join('%s=%.1f' % (a.replace('preprocess','pre').replace('render_','R').replace('radix_','').replace('tile_','').replace('_entries','').replace('activate','act').replace('_views',''), b) for a, b in k.items()))"
  done
done
