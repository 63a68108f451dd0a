#!/usr/bin/env python3
"""Print the start/end times of consecutive kernels from a rocprofv3 kernel trace (CSV):
   timeline.py <trace.csv> [kernel-name substring to anchor on] [number of rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_distance_topk<0>"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s q%-2s lds %-6s vgpr %-4s %9.1f %9.1f %7.1f" % (r["Kernel_Name"][:44], r["Queue_Id"], r["LDS_Block_Size"],
          r["VGPR_Count"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
