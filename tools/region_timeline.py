#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (CSV) of `bench.py --steps K`, print the last K launches of the
distance kernel (the timed region) with the gaps between them and the tail kernels after the last:
   region_timeline.py <trace.csv> [K=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_distance_topk<0>" in r["Kernel_Name"]]
first = idx[-K]
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first - 6:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void sfm::", "").replace("sfm::", "")
    gap = "" if prev_end is None or "k_distance" not in name else "gap %6.1f" % ((s - prev_end) / 1e3)
    print("%-28s q%-2s %9.1f %9.1f %7.1f %s" % (name[:28], r["Queue_Id"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
    if "k_distance" in name:
        prev_end = e
