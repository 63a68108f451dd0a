"""Timeline of the last N GPU activities (kernels and copies) of a rocprofv3 trace directory
(--kernel-trace --memory-copy-trace, csv): start relative to the first of them, duration, idle gap.
    python tools/trace_timeline.py <dir> [N=60]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:30]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:24]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = rows[0][0]
prev = rows[0][0]
for s, e, name in rows:
    print("%9.1f  %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0.0, (s - prev) / 1e3), name))
    prev = max(prev, e)
