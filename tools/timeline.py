#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: span, GPU-busy union, idle gaps, and the
per-kernel share of the busy time.  Usage: timeline.py <dir with *_kernel_trace.csv> [steps_to_skip]"""
import csv, glob, os, sys
from prof_summarize import short

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
rows.sort()
# a step begins at every 4th preprocess_fwd burst: find starts of bursts (gap since last preprocess_fwd > 200 us)
starts = []
last = -10**18
for s, e, n in rows:
    if n == "preprocess_fwd":
        if s - last > 200_000:
            starts.append(s)
        last = s
want = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # preprocess_fwd launches per step (views per call)
cands = []
for a, b in zip(starts[:-1], starts[1:]):
    pre = [r for r in rows if a <= r[0] < b and r[2] == "preprocess_fwd"]
    # one multi-view launch per step (batched call) or `want` single-view launches
    if (want > 1 and len(pre) == 1 and (pre[0][1] - pre[0][0]) > 15_000) or len(pre) == want:
        cands.append((a, b))
t0, t1 = cands[len(cands) // 2]
print(f"{len(cands)} steps with {want} views found; showing the middle one")
step = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
span = (t1 - t0) / 1e3
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(((s - cur_e) / 1e3, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step span {span:.1f} us, GPU busy (union) {busy/1e3:.1f} us, kernels {len(step)}")
print("largest idle gaps (us, next kernel):", sorted(gaps, reverse=True)[:8])
tot = {}
for s, e, n in step:
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
for n, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"  {n:28s} {v:8.1f} us summed")
print("sequence (name, start us, dur us):")
for s_, e_, n in step:
    print(f"   {n[:40]:40s} {(s_ - t0) / 1e3:8.1f} {(e_ - s_) / 1e3:7.1f}")
print("first/last kernels:", [(n, round((s - t0) / 1e3, 1), round((e - t0) / 1e3, 1)) for s, e, n in step[:3]], "...",
      [(n, round((s - t0) / 1e3, 1), round((e - t0) / 1e3, 1)) for s, e, n in step[-3:]])
if os.environ.get("TIMELINE_FULL"):
    rws = [r for r in csv.DictReader(open(f)) if t0 <= int(r["Start_Timestamp"]) < t1]
    for r in rws:
        n = r["Kernel_Name"]
        if "gsr" in n or "anonymous" in n:
            continue
        print("   FULL", n[:230], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
