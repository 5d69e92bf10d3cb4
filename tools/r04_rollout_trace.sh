#!/bin/bash
# per-kernel trace of the graphed rollout step at configs[4] size (500 k Gaussians, 100 bones): which kernels make up its ~0.6 ms
O=$PWD/gpurun_out/r04q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o rollout -- python $GRAFT_REPO_ROOT/tools/rollout_graph_loop.py > $O/run.log 2>&1
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" > $O/rollout_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel-time total %.1f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:45]:
    print("%-110s calls %6s  avg %8.1f us  total %7.2f ms  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -3 $O/run.log; head -50 $O/rollout_kernel_stats.txt
