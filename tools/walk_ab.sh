#!/bin/bash
# Runs on the GPU box: kernel statistics of the bucket walk for library variants built by
# tools/build_variant.sh with -DSFM_ABLATE_WALK_UPTO=1..3 (a1: first-image loads and offsets only;
# a2: + staging of the second image's run; a3: + the scan without its hits; a4: everything).
cd /tmp && export TMPDIR=/tmp
for v in a1 a2 a3 a4; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/cprof_$v
  SFM_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/ab/$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cprof_$v -- python3 $GRAFT_REPO_ROOT/tools/cascade_bench.py --cpu-pairs 0 > $GRAFT_REPO_ROOT/gpurun_out/cprof_$v.log 2>&1 || exit 1
  echo $v; grep -E "k_ch_walk_count" $GRAFT_REPO_ROOT/gpurun_out/cprof_$v/*/*_kernel_stats.csv | cut -d, -f3-8
done
