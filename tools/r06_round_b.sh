#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_b; mkdir -p $O
python -m pytest $R/tests/test_dropin_gpu.py $R/tests/test_bwd_pc_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
printf "301 123\n" | bash $R/tools/r06_diag_cases.sh
python $R/tools/r06_v1_graph.py > $O/v1_graph.txt 2>&1; cat $O/v1_graph.txt
for v in 1 2 4; do
  echo "## --views $v --no-optimizer" >> $O/fwd_prio.txt
  bash $R/tools/ab_env.sh "--views $v --no-optimizer" "GSR_FWD_PRIO_FRAC=0" "GSR_FWD_PRIO_FRAC=16" "GSR_FWD_PRIO_FRAC=48" "GSR_FWD_PRIO_FRAC=128" >> $O/fwd_prio.txt 2>&1
done
cat $O/fwd_prio.txt
