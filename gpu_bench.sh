#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u bench.py --steps 300 --warmup 30 2>&1 | tee gpurun_out/bench.log | tail -5 &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo rc=$?;
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*stats*" | head; 
