#!/bin/bash
# Round 5: occupancy variants of the producer / consumer backward ("lib:WG per CU" pairs), alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
VIEWS=${VIEWS:-4}
{ for round in 1 2; do for pair in "$@"; do lib=${pair%%:*}; w=${pair##*:}; pc=1; [ "$w" = "old" ] && pc=0
  GSR_BWD_PC=$pc GSR_BWD_PC_WG_PER_CU=$w GSR_HIP_LIB=$R/gs-dynamics_amd/csrc/$lib timeout 300 python $R/bench.py --views $VIEWS --no-optimizer --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['per_kernel_us_per_launch']
print('%-28s step %.1f us | Rbwd=%.1f Rfwd=%.1f' % ('$pair', r['step_us'], k['render_bwd'], k['render_fwd']))"
done; done; } > $O/r05_pc_variants.txt 2>&1
