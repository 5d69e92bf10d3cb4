#!/bin/bash
# the one-launch propagation network: dynamics GPU tests, then the rollout step traced per kernel with and without it
O=$PWD/gpurun_out/r04z; mkdir -p $O
( timeout 900 python -m pytest tests/test_dynamics_gpu.py -m gpu -x -q -s 2>&1 | tail -15 ) > $O/pytest.log 2>&1; cat $O/pytest.log
for cfg in "GSDYN_GNN_SPLIT=1" "GSDYN_GNN_SPLIT=0" "GSDYN_GNN_FUSED=1" "GSDYN_GNN_SPLIT=1"; do
  echo "$cfg: $(env $cfg timeout 300 python tools/rollout_graph_loop.py 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o rollout -- python $GRAFT_REPO_ROOT/tools/rollout_graph_loop.py > $O/run.log 2>&1
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" > $O/rollout_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel-time total %.1f ms over %d kernels (205 steps + set-up)" % (tot / 1e6, len(rows)))
for r in rows[:30]:
    print("%-100s calls %6s  avg %8.1f us  total %7.2f ms  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "graphed rollout" $O/run.log; head -16 $O/rollout_kernel_stats.txt | cut -c1-190
