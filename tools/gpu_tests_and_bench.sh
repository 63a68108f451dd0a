#!/bin/bash
# parity tests, then bench, then a kernel trace of a short bench run
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 180 --timeout-method thread 2>&1 | tee gpurun_out/tests.log | tail -15 &&
timeout -k 10 600 python -u bench.py ${BENCH_ARGS:---steps 300 --warmup 30} 2>&1 | tee gpurun_out/bench.log | tail -3 &&
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-cascade --no-batched > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 &&
cat $GRAFT_REPO_ROOT/gpurun_out/prof/*/*_kernel_stats.csv
