#!/bin/bash
# Runs on the GPU box: cascade parity tests, then the cascade bench's kernel statistics for the
# library as built and for the variants named on the command line (tools/ab/<name>.so).
set -o pipefail
R=$GRAFT_REPO_ROOT
[ -z "$SKIP_TESTS" ] && timeout -k 10 400 python -u -m pytest $R/tests/test_gpu_cascade.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for v in lib "$@"; do
  rm -rf $R/gpurun_out/cab_$v
  lib=$R/sfm_amd/lib/libsfm_hip.so; [ $v != lib ] && lib=$R/tools/ab/$v.so
  SFM_HIP_LIBRARY=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cab_$v -- python3 $R/tools/cascade_bench.py --cpu-pairs 0 > $R/gpurun_out/cab_$v.log 2>&1 || { tail -5 $R/gpurun_out/cab_$v.log; exit 1; }
  echo "== $v: $(grep -o '"pairs_per_s": [0-9.]*' $R/gpurun_out/cab_$v.log)"
  python3 $R/tools/kstats.py $R/gpurun_out/cab_$v/*/*_kernel_stats.csv k_ch_ | head -4
done
