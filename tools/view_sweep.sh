#!/bin/bash
# 1-GPU latency table of the step for V = 1, 2, 4, 8 views (VERDICT r01 next #1b): render fwd+bwd (direct calls) + Adam.
# Output: gpurun_out/view_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
: > $O/view_sweep.txt
for v in 1 2 4 8; do
  timeout 300 python $R/bench.py --views $v --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['per_kernel_us_per_launch']
print('V=$v  step %.1f us (event median %.1f)  %.0f Mpix/s  busy %.0f us | ' % (r['step_us'], 1e3*d['ms_per_step_event_median'], d['value'], r['gsr_kernels_busy_us_per_step']) + ' '.join('%s=%.1f' % (a, b) for a, b in k.items()))" >> $O/view_sweep.txt
done
cat $O/view_sweep.txt
