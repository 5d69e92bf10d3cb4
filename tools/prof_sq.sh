#!/bin/bash
# SQ counters of the blend kernels (run on the GPU box): where the wave cycles go.  Two --pmc passes (8 SQ counters each),
# kernel trace only (no sys/hip/hsa trace domains).  Output: gpurun_out/sq_render.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for set in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set --kernel-include-regex "render_" -d $O/prof_sq$i -o run -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_sq$i.log 2>&1
done
cd $R
{ echo "# rocprofv3 --pmc, bench.py default call pattern (one launch = 4 views); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles";
  python tools/prof_summarize.py pmc $O/prof_sq1; python tools/prof_summarize.py pmc $O/prof_sq2; } > $O/sq_render.txt
python - <<'PY' >> $O/sq_render.txt
import csv, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("prof_sq1", "prof_sq2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = "render_fwd" if "render_fwd" in r["Kernel_Name"] else "render_bwd" if "render_bwd" in r["Kernel_Name"] else None
            if k:
                c = acc[k][r["Counter_Name"]]; c[0] += 1; c[1] += float(r["Counter_Value"])
import json
json.dump({k: {n: t / max(c, 1) for n, (c, t) in v.items()} for k, v in acc.items()} | {"views_per_launch": 4,
          "source": "rocprofv3 --pmc SQ_* (two passes), tools/prof_sq.sh; means per dispatch; *_CYCLES / ACTIVE / WAIT in quad-cycles"},
          open(os.path.join(O, "sq_counters.json"), "w"), indent=1)
print("# derived (means per dispatch)")
for k, v in acc.items():
    m = {n: t / max(c, 1) for n, (c, t) in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: VALU-active share of wave cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}, LDS-active {m.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}, "
          f"any-inst active {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}, parked (waitcnt/barrier) {m.get('SQ_WAIT_ANY', 0) / wc:.3f}, "
          f"issue-stalled {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}; VALU instr per wave {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_WAVES', 1), 1):.0f}, "
          f"LDS bank-conflict share of LDS cycles {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.4f}")
PY
rm -rf $O/prof_sq1 $O/prof_sq2
cat $O/sq_render.txt
