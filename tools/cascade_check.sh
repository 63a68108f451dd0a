#!/bin/bash
# cascade-hash path: parity tests, throughput line, per-kernel times
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/cascade_check; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_cascade.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -4 &&
timeout -k 10 300 python -u $R/tools/cascade_bench.py --cpu-pairs 0 2>$OUT/bench.err | tee $OUT/cascade_bench.json | cut -c1-700 &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/cascade_bench.py --cpu-pairs 0 > $OUT/stats.log 2>&1 &&
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv && cut -d, -f1-4,6,7 $OUT/kernel_stats.csv | cut -c1-160
