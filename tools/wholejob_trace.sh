#!/bin/bash
# Kernel + HIP API trace of bench.py's whole-job leg alone (100 images): launch gaps, and what the uploads cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wj_trace; rm -rf $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-engine --no-cascade --no-batched > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 $R/tools/wholejob_timeline.py $O/*/*kernel_trace.csv
python3 - $O/*/*kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if "k_distance_topk<0>" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 1000 * 256]
print("durations (ms) and gaps (ms) of the big launches in order:")
prev = None
for r in big:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%.2f(%.2f)" % ((e - s) / 1e6, 0 if prev is None else (s - prev) / 1e6), end=" ")
    prev = e
print()
PY
grep -E "hipMalloc|hipMemcpyAsync|hipFree|hipMemsetAsync|hipStreamSynchronize" $O/*/*hip_api_stats.csv
