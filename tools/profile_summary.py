#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs that tools/profile.sh left under gpurun_out/prof_<tag>/ into one text table
(per kernel: calls, total/avg duration, PMC counter sums).  Usage: profile_summary.py gpurun_out/prof_r01 > profiles/r01_summary.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:48]


def find(pass_name, suffix):
    hits = glob.glob(os.path.join(root, pass_name, "**", f"*{suffix}"), recursive=True)
    return hits[0] if hits else None


print(f"# rocprofv3 summary of {root}")
f = find("trace", "kernel_stats.csv")
if f:
    print("\n## kernel-trace --stats (kernel_stats.csv)")
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':34s} {'calls':>7s} {'total_ms':>12s} {'avg_us':>12s} {'pct':>7s}")
    for r in rows:
        name = short(r.get("Name", ""))
        if "k_" not in name:
            continue
        print(f"{name:34s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:12.3f} {float(r['AverageNs']) / 1e3:12.2f} {float(r['Percentage']):7.2f}")
f = find("trace", "kernel_trace.csv")
if f:
    agg = defaultdict(lambda: [0, 0.0, 0, 0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a[2] = int(r.get("VGPR_Count", 0) or 0)
        a[3] = int(r.get("SGPR_Count", 0) or 0)
        a[4] = int(r.get("LDS_Block_Size", 0) or 0)
    print("\n## kernel_trace.csv (per kernel: dispatches, total ms, VGPR, SGPR, LDS)")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if k.startswith("k_"):
            print(f"{k:34s} {a[0]:7d} {a[1]:12.3f} ms  avg {a[1] / a[0] * 1e3:10.2f} us  vgpr {a[2]:4d} sgpr {a[3]:4d} lds {a[4]:6d}")

for p in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(p):
        continue
    f = find(os.path.basename(p), "counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    if "--per-dispatch" in sys.argv:
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            per[(int(r["Dispatch_Id"]), short(r["Kernel_Name"]))][r["Counter_Name"]] = float(r["Counter_Value"])
        print(f"\n## {os.path.basename(p)} per dispatch")
        for (d, k), v in sorted(per.items()):
            if k.startswith("k_"):
                print(f"{d:5d} {k:26s} " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
        continue
    print(f"\n## {os.path.basename(p)} (counter sums over all dispatches of the kernel)")
    for k in sorted(agg, key=lambda kk: -sum(agg[kk].values())):
        if not k.startswith("k_"):
            continue
        print(f"{k:34s} dispatches {len(calls[k]):6d}  " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(agg[k].items())))
