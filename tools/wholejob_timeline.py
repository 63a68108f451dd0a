#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of a run with 64-pair calls: the gaps between consecutive launches of the
distance kernel with more than 1000 workgroups, their durations, and what ran in the gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if "k_distance_topk<0>" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 1000 * 256]
gaps, durs = [], []
for a, b in zip(big, big[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
gaps.sort()
n = len(gaps)
print("launches %d  duration mean %.1f us  gap median %.1f  p90 %.1f  max %.1f  mean %.1f" %
      (len(big), sum(durs) / n, gaps[n // 2], gaps[n * 9 // 10], gaps[-1], sum(gaps) / n))
for name in ("k_rerank<64>", "k_cross_compact<1024>", "copyBuffer"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    if d:
        print("%-24s n %4d mean %.1f us" % (name, len(d), sum(d) / len(d)))
