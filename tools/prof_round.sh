#!/bin/bash
# Round profile set on the GPU box (bench.py default = the 8-view step): kernel stats, HBM traffic of the blend kernels (calibrated),
# SQ counters.  Outputs under gpurun_out/ (copy to profiles/ with the round prefix):
#   kernel_stats.txt, pmc_traffic.json, sq_counters.json, sq_render.txt, marker_ranges.txt
# --pmc passes use --kernel-trace only (no sys/hip/hsa/marker trace domains next to counters).
# usage: tools/prof_round.sh [VIEWS]   (8 = bench.py default; 4 = --config 3, BASELINE.json's own 4 x 800^2 configuration)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
VIEWS=${1:-8}; export GSR_PROF_VIEWS=$VIEWS
WL="--views $VIEWS"; [ "$VIEWS" = "4" ] && WL="--config 3"
BENCH="python $R/bench.py $WL --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o run -- python $R/bench.py $WL --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_stats.log 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/prof_$c
  rocprofv3 --kernel-trace --output-format csv --pmc $c --kernel-include-regex "render_" -d $O/prof_$c -o run -- $BENCH > $O/prof_$c.log 2>&1 || true
done
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for set in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $O/prof_sq$i
  rocprofv3 --kernel-trace --output-format csv --pmc $set --kernel-include-regex "render_" -d $O/prof_sq$i -o run -- $BENCH > $O/prof_sq$i.log 2>&1 || true
done
rm -rf $O/prof_marker && rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/prof_marker -o run -- python $R/bench.py $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_marker.log 2>&1 || true
cd $R
python tools/prof_summarize.py stats $O/prof_stats > $O/kernel_stats_v$VIEWS.txt || true
python - <<'PY'
import csv, glob, json, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()); O = os.path.join(R, "gpurun_out")
P, Npx, V = 100_000, 640_000, int(os.environ.get("GSR_PROF_VIEWS", "8"))
def kname(n):
    return "render_fwd" if "render_fwd" in n else "render_bwd" if "render_bwd" in n else None
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("prof_FETCH_SIZE", "prof_WRITE_SIZE", "prof_sq1", "prof_sq2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                c = acc[k][r["Counter_Name"]]; c[0] += 1; c[1] += float(r["Counter_Value"])
mean = {k: {n: t / max(c, 1) for n, (c, t) in v.items()} for k, v in acc.items()}
# entries per view of the benchmark scene (printed by bench.py; constant for SynthScene-v1)
try:
    D = json.loads(open(os.path.join(O, "prof_stats.log")).read().strip().splitlines()[-1])["config"]["num_rendered_per_view"]
except Exception:
    D = 414543.0
cal = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_round.sh), corrected with profiles/r02_traffic_calibration.json: "
                 "coalesced reads are counted at 1/2, 64-byte record gathers at 1, scattered 36-byte stores cost 1.81 fabric bytes per byte written",
       "views_per_launch": V, "entries_per_view": D}
for k, m in mean.items():
    f_raw, w_raw = m.get("FETCH_SIZE", 0.0) * 1024.0, m.get("WRITE_SIZE", 0.0) * 1024.0
    stream_rd = V * (4.0 * D + (20.0 * Npx if k == "render_bwd" else 0.0))        # point_list (+ the per-pixel inputs of the backward)
    fetched = f_raw + 0.5 * stream_rd                                              # the streaming part was counted at one half
    written = V * (36.0 * D if k == "render_bwd" else 24.0 * Npx)                  # bytes actually stored
    cal[k] = {"FETCH_SIZE_KiB_raw": f_raw / 1024.0, "WRITE_SIZE_KiB_raw": w_raw / 1024.0, "fabric_bytes_per_launch": fetched + w_raw,
              "hbm_bytes_per_launch": fetched + written,
              "note": "hbm_bytes = bytes fetched (calibrated) + bytes stored; fabric_bytes counts the partial-sector cost of the scattered record stores too"}
json.dump(cal, open(os.path.join(O, f"pmc_traffic_v{V}.json"), "w"), indent=1)
sq = {k: {n: v for n, v in m.items() if n.startswith("SQ_")} for k, m in mean.items()}
sq.update({"views_per_launch": V, "clock_hz": 2.2e9,
           "source": "rocprofv3 --pmc SQ_* (two passes), tools/prof_round.sh; means per dispatch; *_CYCLES / ACTIVE / WAIT in quad-cycles"})
json.dump(sq, open(os.path.join(O, f"sq_counters_v{V}.json"), "w"), indent=1)
with open(os.path.join(O, f"sq_render_v{V}.txt"), "w") as fh:
    fh.write(f"# rocprofv3 --pmc, bench.py (one launch = {V} views); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles\n")
    for k, m in mean.items():
        for n in sorted(m):
            fh.write(f"{k:12s} {n:24s} {m[n]:16.1f}\n")
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        fh.write(f"{k}: VALU-active share of wave cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}, parked {m.get('SQ_WAIT_ANY', 0) / wc:.3f}, "
                 f"issue-stalled {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, LDS array busy cycles {m.get('SQ_LDS_IDX_ACTIVE', 0):.0f}\n")
print(json.dumps(cal, indent=1)); print(open(os.path.join(O, f"sq_render_v{V}.txt")).read())
# marker ranges: what a third-party timeline shows
rows = collections.Counter()
for f in glob.glob(os.path.join(O, "prof_marker", "**", "*marker_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r.get("Function", r.get("Name", "?"))] += 1
open(os.path.join(O, "marker_ranges.txt"), "w").write("# roctx ranges seen by rocprofv3 --marker-trace (count over 3 steps)\n" + "\n".join(f"{n:40s} {c}" for n, c in rows.most_common()))
print(open(os.path.join(O, "marker_ranges.txt")).read()[:1500])
PY
rm -rf $O/prof_FETCH_SIZE $O/prof_WRITE_SIZE $O/prof_sq1 $O/prof_sq2 $O/prof_stats $O/prof_marker
cat $O/kernel_stats_v$VIEWS.txt | head -30
