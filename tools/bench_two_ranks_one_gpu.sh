#!/bin/bash
# two (or RANKS) ranks of the multi-GPU bench sharing the one GPU of the box, collectives over gloo
set -o pipefail
mkdir -p gpurun_out
SFM_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${RANKS:-2} --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus ${RANKS:-2} --steps 20 --warmup 5 --set-images ${SET_IMAGES:-60} 2>&1 | tee gpurun_out/bench_ranks.log | tail -2
