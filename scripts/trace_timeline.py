#!/usr/bin/env python3
"""Launch timeline of the LAST window of a rocprofv3 --kernel-trace run: every kernel from the last launch whose name contains
<marker> on, with start offset, duration, gap to the previous end on the same queue, and queue id.
usage: trace_timeline.py <rocprof output dir> <marker>"""
import csv, glob, os, sys
d, marker = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if marker in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
ends = {}
for r in rows[last:]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - ends[q]) / 1e3 if q in ends else 0.0
    ends[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{q}  {r['Kernel_Name'][:70]}")
print(f"window: {(max(int(r['End_Timestamp']) for r in rows[last:]) - t0) / 1e3:.1f} us")
