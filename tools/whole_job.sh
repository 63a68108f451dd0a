#!/bin/bash
# whole-job numbers: the C++ class on 100 x 8192 (both engines), then bench.py's default run
set -o pipefail
mkdir -p gpurun_out
for e in 0 1; do timeout -k 10 200 sfm_amd/lib/match_set_bench 100 8192 0.5 0 $e || exit 1; done 2>&1 | tee gpurun_out/match_set.log &&
SFM_MATCH_BENCH_REPEAT=2 timeout -k 10 200 sfm_amd/lib/match_set_bench 100 8192 0.5 0 0 2>&1 | tee -a gpurun_out/match_set.log &&
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q -k "wait_outputs or reuse_output or in_flight" --timeout 120 --timeout-method thread 2>&1 | tail -3 &&
timeout -k 10 600 python -u bench.py --steps 300 --warmup 30 --no-cascade 2>&1 | tee gpurun_out/bench.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'whole_job',{k:d['whole_job'][k] for k in ('uploads_s','match_s','pairs_per_s_matching','value')}, 'roof', d['whole_job']['roofline']['frac'])"
