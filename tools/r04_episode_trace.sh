#!/bin/bash
# per-kernel totals of the configs[4] episode (rollout + renders): where the render's 1.67 ms per frame go beyond the 0.90 ms of the frame loop
O=$PWD/gpurun_out/r04ep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o ep -- python $GRAFT_REPO_ROOT/bench.py --config 5 --with-rollout --steps 30 > $O/run.log 2>&1
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" > $O/episode_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel-time total %.1f ms over %d kernels (2 episodes of 30 frames: warm-up + timed)" % (tot / 1e6, len(rows)))
for r in rows[:40]:
    print("%-90s calls %6s  avg %8.1f us  total %7.2f ms  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -2 $O/run.log | cut -c1-300; head -45 $O/episode_kernel_stats.txt | cut -c1-170
