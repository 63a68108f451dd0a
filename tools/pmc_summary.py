#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, the mean of every counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name in sorted(acc):
        print(name)
        for c in sorted(acc[name]):
            v = acc[name][c]
            print("    %-32s n=%-5d mean=%.6g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
