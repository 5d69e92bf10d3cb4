#!/bin/bash
# per-kernel durations of the long-list scene (tools/r05_long_list_scene.py), with and without the cluster
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in both base; do
  rm -rf /tmp/prof_ll_$arm
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ll_$arm -o run -- python $R/tools/r05_long_list_scene.py $arm > /tmp/prof_ll_$arm.log 2>&1 || true
  python - <<PY > $O/r05_long_list_$arm.txt
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("/tmp/prof_ll_$arm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if "gsr_" in k:
        print("%-112s calls %4d  avg %8.1f us  max %8.1f" % (k, len(v), sum(v) / len(v), max(v)))
PY
done
