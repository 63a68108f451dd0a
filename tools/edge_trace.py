"""Where the edges of a short timed region go: reads a rocprofv3 kernel trace of
    python3 bench.py --steps K --warmup W --no-cpu-baseline --no-other-engine --no-whole-job --no-batched --no-cascade --no-geometry
and prints, for the last K launches of the distance kernel (the timed region) and the kernels between
them, start (relative), duration and the idle gap before each.
    python tools/edge_trace.py <dir with *_kernel_trace.csv> K"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
k = int(sys.argv[2])
dist = [i for i, r in enumerate(rows) if "k_distance_topk" in r[2]]
first = dist[-k]
t0 = rows[first][0]
prev_end = rows[first - 1][1]
print("idle before the region's first distance launch: %.1f us" % ((rows[first][0] - prev_end) / 1e3))
for s, e, name in rows[first:]:
    short = name.split("(")[0].split("::")[-1][:28]
    print("%9.1f  %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if s > prev_end else 0.0, short))
    prev_end = max(prev_end, e)
print("region, first start to last end: %.1f us" % ((max(r[1] for r in rows[first:]) - t0) / 1e3))
