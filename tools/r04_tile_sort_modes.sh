#!/bin/bash
# per-launch times of the per-tile sort's modes: configs[4] frame loop (static scene) and episode (deforming scene), long tickets on the 2048- / 4096-entry LDS block
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04ts; mkdir -p $O
for what in "static:--config 5 --no-cpu-baseline" "episode:--config 5 --with-rollout --steps 30"; do
  name=${what%%:*}; args=${what#*:}
  for w in 2048 4096; do
    rm -rf $O/t; GSR_LONG_SORT=$w timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o ep -- python $GRAFT_REPO_ROOT/bench.py $args > $O/run.log 2>&1
    f=$(ls $O/t/*kernel_stats.csv $O/t/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $name GSR_LONG_SORT=$w"
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tile_sort' in r['Name'] or 'render_fwd' in r['Name'] or 'bin_emit' in r['Name']:
        print("  %-72s calls %5s avg %8.1f us total %7.2f ms" % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  done
done
