#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches / memory copies of a rocprofv3 --kernel-trace --memory-copy-trace run:
   python tools/trace_timeline.py <dir with *_kernel_trace.csv, *_memory_copy_trace.csv> [N]"""
import csv, glob, os, sys
d = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q" + r.get("Queue_Id", "?") + " " + r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Name", "")))
rows.sort()
rows = rows[-N:]
t0 = rows[0][0]
for s, e, n in rows:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n}")
