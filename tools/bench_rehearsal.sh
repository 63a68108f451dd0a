#!/bin/bash
# the multi-rank bench path on one GPU (one rank, RCCL group, sharded job, gather), then the default run
set -o pipefail
mkdir -p gpurun_out
SFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 400 python -u bench.py --gpus 1 --steps 20 --warmup 5 --set-images ${SET_IMAGES:-60} 2>&1 | tee gpurun_out/bench_dist1.log | tail -2 &&
timeout -k 10 600 python -u bench.py ${BENCH_ARGS:---steps 300 --warmup 30} 2>&1 | tee gpurun_out/bench.log | tail -2
