#!/bin/bash
# per-kernel averages over a default bench.py run (headline + scale_n1 + extras): a screen for kernels that are slower than their work explains
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04bt; mkdir -p $O; rm -rf $O/t
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/run.log 2>&1
f=$(ls $O/t/*kernel_stats.csv $O/t/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" > $O/bench_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel-time total %.1f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:60]:
    print("%-84s calls %6s  avg %8.1f us  total %7.2f ms  %5.1f %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -62 $O/bench_kernel_stats.txt | cut -c1-160
