#!/bin/bash
# A/B of environment settings on the UNCHANGED-caller numbers of bench.py's extras (separate GaussianRasterizer calls): tools/ab_dropin.sh "VAR=a" "VAR=b" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout 600 python $R/bench.py --views 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']; m=e['train_iteration_one_camera']['ms']
print('%-24s separate_calls t0 %.3f t>0 %.3f ms/camera; train iteration reference_shape t0 %.3f t>0 %.3f; predict frame reference calls %.3f (fused %.3f); cfg2 fwd %.3f' % ('$cfg', e['getloss_step_t0']['separate_calls']['ms_per_view'], e['getloss_step']['separate_calls']['ms_per_view'], m['t0 reference_shape'], m['t>0 reference_shape'], e['predict_frame_4cams']['ms_per_frame_reference_calls'], e['predict_frame_4cams']['ms_per_frame'], e['forward_only_cfg2']['ms_per_view']))"
  done
done
