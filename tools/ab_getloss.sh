#!/bin/bash
# A/B of library builds on the get_loss-shaped step (fused pairs): tools/ab_getloss.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for lib in "$@"; do
    GSR_HIP_LIB=$R/gs-dynamics_amd/csrc/$lib timeout 600 python $R/bench.py --views 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']
print('%-22s getloss t0 %.3f ms, t>0 %.3f ms; one-camera iteration t0 %.3f / t>0 %.3f; predict frame %.3f; cfg2 fwd %.3f' % ('$lib', e['getloss_step_t0']['ms_per_step'], e['getloss_step']['ms_per_step'], e['train_iteration_one_camera']['ms']['t0 fused'], e['train_iteration_one_camera']['ms']['t>0 fused'], e['predict_frame_4cams']['ms_per_frame'], e['forward_only_cfg2']['ms_per_view']))"
  done
done
