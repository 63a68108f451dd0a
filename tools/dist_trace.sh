#!/bin/bash
# Kernel trace of the sharded bench on one rank: durations of the distance launches of the last steps and the gaps between them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dist_trace; rm -rf $O
cd /tmp && export TMPDIR=/tmp
SFM_BENCH_BACKEND=${BACKEND:-nccl} SFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --gpus 1 --steps 40 --warmup 5 --set-images 60 --no-whole-job --no-replication-check $EXTRA > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 - $O/*/*kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [r for r in rows if "k_distance_topk" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 4e6]
prev = None
for r in d[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%.3f(gap %.3f)" % ((e - s) / 1e6, 0 if prev is None else (s - prev) / 1e6), end=" ")
    prev = e
print()
PY
