#!/usr/bin/env python3
"""Kernel timeline of the LAST frame of a rocprofv3 --kernel-trace run: start (us since the first listed dispatch),
duration (us), kernel, grid.   python tools/trace_timeline.py <rocprof dir> [n_last_dispatches]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", r["Kernel_Name"])
    name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name:28s} grid {r.get('Grid_Size_X', '?')}")
    prev_end = max(prev_end, e)
