#!/bin/bash
# Runs on the GPU box: vector, scalar and LDS instruction counts of k_ch_walk_count for the library as
# built and for the variants cut off after each phase (tools/ab/walk_upto<k>.so, -DSFM_ABLATE_WALK_UPTO=k).
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in lib "$@"; do
  rm -rf $R/gpurun_out/wi_$v
  lib=$R/sfm_amd/lib/libsfm_hip.so; [ $v != lib ] && lib=$R/tools/ab/$v.so
  SFM_HIP_LIBRARY=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/wi_$v -- python3 $R/tools/cascade_bench.py --cpu-pairs 0 --reps 2 > $R/gpurun_out/wi_$v.log 2>&1 || { tail -5 $R/gpurun_out/wi_$v.log; exit 1; }
  echo "== $v"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/wi_$v /dev/null 2>/dev/null | grep -A5 "k_ch_walk_count" | head -6
done
