#!/bin/bash
# per-kernel durations of the 3 M-Gaussian / four 1080p views step (tools/r05_big_scene.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_big
BIG_ONLY_FIRST=1 timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_big -o run -- python $R/tools/r05_big_scene.py > /tmp/prof_big.log 2>&1 || true
python - <<PY > $O/r05_big_scene_kernels.txt
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("/tmp/prof_big/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if "gsr_" in k:
        print("%-112s calls %4d  avg %8.1f us  max %8.1f" % (k, len(v), sum(v) / len(v), max(v)))
PY
tail -3 /tmp/prof_big.log >> $O/r05_big_scene_kernels.txt
