#!/usr/bin/env python
"""Summarise rocprofv3 output directories into small text files that can be committed under profiles/.

  python tools/prof_summarize.py stats   <dir> > profiles/rNN_kernel_stats.txt     (--kernel-trace --stats run)
  python tools/prof_summarize.py pmc     <dir> > profiles/rNN_pmc_<set>.txt        (--pmc run: mean counter per kernel)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    for tag in ("render_fwd", "render_bwd", "preprocess_fwd", "preprocess_bwd", "emit_entries", "radix_hist", "radix_scatter",
                "tile_ranges_order", "tile_order", "tile_sort", "scan_exclusive", "mark_visible", "bin_count", "bin_scan_order", "bin_scan", "bin_emit",
                "bin_colprefix", "adam_step"):
        if tag in name:
            return tag
    name = name.replace("void ", "").replace("at::native::", "")
    return name[:60]


def stats(d):
    files = find(d, "*kernel_stats.csv")
    if not files:
        # fall back: aggregate the kernel trace
        rows = defaultdict(lambda: [0, 0.0])
        for f in find(d, "*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                k = short(r["Kernel_Name"])
                rows[k][0] += 1
                rows[k][1] += dur
        tot = sum(v[1] for v in rows.values()) or 1
        print(f"{'kernel':40s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
        for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:40s} {n:8d} {t:12.1f} {t / n:10.2f} {100 * t / tot:6.2f}")
        return
    for f in files:
        print("#", os.path.basename(f))
        rd = list(csv.DictReader(open(f)))
        print(f"{'kernel':40s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
        for r in rd:
            print(f"{short(r['Name']):40s} {int(r['Calls']):8d} {float(r['TotalDurationNs']) / 1e3:12.1f} "
                  f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")


def pmc(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = acc[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    for k in sorted(acc):
        print(k)
        for cn in sorted(acc[k]):
            n, t = acc[k][cn]
            print(f"    {cn:32s} mean/dispatch {t / n:16.1f}   dispatches {n}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
