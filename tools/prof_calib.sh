#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE on known byte counts in the blend kernels' access patterns (run on the GPU box).
# Output: gpurun_out/traffic_calibration.json  (copy to profiles/).  Separate --pmc passes, kernel trace only.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
B=$R/tools/micro/traffic_calib
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/traffic_calib.hip -o $B
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/calib_$c
  rocprofv3 --kernel-trace --output-format csv --pmc $c --kernel-include-regex "calib_" -d $O/calib_$c -o run -- $B > $O/calib_$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, os, re
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
known = []
for line in open(os.path.join(O, "calib_FETCH_SIZE.log")):
    m = re.match(r"KNOWN (\S+) (\d) read (\d+) write (\d+)(?: index (\d+))?", line)
    if m:
        known.append(dict(kernel=m.group(1), nth=int(m.group(2)), read=int(m.group(3)), write=int(m.group(4)), index=int(m.group(5) or 0)))
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(O, "calib_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), re.sub(r"\(.*", "", r["Kernel_Name"]).split()[-1], float(r["Counter_Value"])))
    rows.sort()
    seen = {}
    for _, k, v in rows:
        n = seen.get(k, 0); seen[k] = n + 1
        vals[(k, n, c)] = v
out = {"source": "tools/prof_calib.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/micro/traffic_calib.hip; counters are KiB",
       "rows": []}
for kn in known:
    f = vals.get((kn["kernel"], kn["nth"], "FETCH_SIZE")); w = vals.get((kn["kernel"], kn["nth"], "WRITE_SIZE"))
    row = dict(kn, FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w)
    rd = kn["read"] + kn["index"]
    if f is not None and rd:
        row["fetch_counter_bytes_per_known_read_byte"] = f * 1024.0 / rd
    if w is not None and kn["write"]:
        row["write_counter_bytes_per_known_written_byte"] = w * 1024.0 / kn["write"]
    out["rows"].append(row)
json.dump(out, open(os.path.join(O, "traffic_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE
