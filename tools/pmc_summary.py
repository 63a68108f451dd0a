#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, the mean of every counter per dispatch."""
import csv
import json
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
    # HBM-side traffic of the distance kernel per launch, per engine (template argument 0 = fp32
    # matrix instructions, 1 = split fp16).  rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB; on
    # gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads (all of this
    # kernel's reads are 16 bytes per lane) count at half their size.
    out = {}
    for name in acc:
        if "k_distance_topk" in name and "FETCH_SIZE" in acc[name] and "WRITE_SIZE" in acc[name]:
            engine = "f16x3" if "<1>" in name else "f32"
            f = acc[name]["FETCH_SIZE"]
            w = acc[name]["WRITE_SIZE"]
            fetch = sum(f) / len(f) * 1024.0
            write = sum(w) / len(w) * 1024.0
            out[engine] = {"kernel": name, "launches_sampled": len(f),
                           "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
                           "fetch_correction": 2.0, "traffic_bytes_per_launch": 2.0 * fetch + write}
    # The cascade-hash path: one batch = one launch each of the kernels below; their traffic summed.
    batch = ["k_ch_walk_count", "k_ch_nearest", "k_ch_take_all", "k_ch_walk_nearest", "k_ch_compact"]
    per = {}
    for name in acc:
        short = name.split("::")[-1]
        if short in batch and "FETCH_SIZE" in acc[name] and "WRITE_SIZE" in acc[name]:
            f, w = acc[name]["FETCH_SIZE"], acc[name]["WRITE_SIZE"]
            per[short] = {"launches_sampled": len(f), "FETCH_SIZE_bytes_raw": sum(f) / len(f) * 1024.0,
                          "WRITE_SIZE_bytes": sum(w) / len(w) * 1024.0}
            per[short]["traffic_bytes_per_launch"] = 2.0 * per[short]["FETCH_SIZE_bytes_raw"] + per[short]["WRITE_SIZE_bytes"]
    if per:
        out["cascade_batch"] = {"kernels": per, "fetch_correction": 2.0,
                                "traffic_bytes_per_batch": sum(v["traffic_bytes_per_launch"] for v in per.values())}
    if out:
        if len(sys.argv) > 2:
            with open(sys.argv[2], "w") as fh:
                json.dump(out, fh, indent=1)
        print("traffic", json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
