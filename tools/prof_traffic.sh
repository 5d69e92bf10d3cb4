#!/bin/bash
# HBM traffic of the blend kernels from the TCC fabric counters (run on the GPU box).
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC has 4 PMC slots: 3 + 2), so two runs.
# MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled here
# (upper bound for our gather-heavy access pattern; WRITE_SIZE is uncalibrated and taken as is); both are in KiB.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c --kernel-include-regex "render_" -d $O/prof_$c -o run -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(O, "prof_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = "render_fwd" if "render_fwd" in r["Kernel_Name"] else "render_bwd" if "render_bwd" in r["Kernel_Name"] else None
            if k and r["Counter_Name"] == c:
                a = acc.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/prof_traffic.sh; FETCH_SIZE x2 per MI355X_MICROARCH.md",
       "views_per_launch": 4, "call_pattern": "bench.py default: one launch per stage for the 4 views of a step"}
for k, v in acc.items():
    f = v.get("FETCH_SIZE", [1, 0.0]); w = v.get("WRITE_SIZE", [1, 0.0])
    fetch_kib, write_kib = f[1] / max(f[0], 1), w[1] / max(w[0], 1)
    out[k] = {"FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
              "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0, "launches": f[0]}
json.dump(out, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE
