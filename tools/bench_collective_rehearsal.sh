#!/bin/bash
# the multi-rank bench with replication by all-gather: one rank over RCCL, then two ranks over gloo on 16k features
set -o pipefail
mkdir -p gpurun_out
SFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29621 timeout -k 10 300 python -u bench.py --gpus 1 --steps 10 --warmup 3 --set-images 40 --replication collective 2>&1 | tee gpurun_out/bench_coll1.log | tail -1 | cut -c1-1500 &&
SFM_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29622 bench.py --gpus 2 --steps 6 --warmup 2 --set-images 16 --set-features 16384 --replication collective 2>&1 | tee gpurun_out/bench_coll2.log | tail -1 | cut -c1-3000
