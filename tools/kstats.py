#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: kstats.py <file> [substring]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print("%-40s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5s%%" % (name[:40], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
