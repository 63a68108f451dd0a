#!/bin/bash
# Runs on the GPU box: the bench line's geometry leg (with its CPU baseline), then rocprofv3 kernel
# statistics of the same command without the CPU part.  Results land under
# gpurun_out/collect_geometry/; copy what is to be kept into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/collect_geometry; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cascade --no-batched --no-whole-job --no-other-engine"
timeout -k 10 300 python -u $R/bench.py $ARGS 2>$OUT/bench.err > $OUT/bench.json || exit 1
python3 -c "import json,sys; d=json.load(open('$OUT/bench.json')); json.dump(d['geometry'], open('$OUT/geometry_bench.json','w'), indent=1); print(json.dumps(d['geometry'])[:600])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS --no-cpu-baseline > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv && python3 $R/tools/kstats.py $OUT/kernel_stats.csv k_geo
