#!/bin/bash
# instruction-cache counters of the distance kernel with and without the tail beside it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$R
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|SQC_INST" | head -20 > $OUT/avail.txt
for v in new notail; do
  SFM_HIP_LIBRARY=$R/tools/ab/$v.so timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/$v -- python3 $R/tools/steprate.py > $OUT/$v.log 2>&1 || { echo "pass $v failed"; tail -5 $OUT/$v.log; }
done
python3 $R/tools/pmc_summary.py $OUT/new | grep -A6 "k_distance_topk<0>"
python3 $R/tools/pmc_summary.py $OUT/notail | grep -A6 "k_distance_topk<0>"
