#!/bin/bash
# Round 5: producer / consumer backward vs the barrier form, same box, alternating (4 views = the metric's step, 1 and 8 views)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
{ for v in ${VIEWS:-4 1 8}; do echo "== views $v"; $R/tools/ab_env.sh "--views $v --no-optimizer --steps 20 --warmup 5" "GSR_BWD_PC=0" "$@"; done; } > $O/r05_pc_ab.txt 2>&1
