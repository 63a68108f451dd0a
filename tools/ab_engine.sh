#!/bin/bash
# The split-fp16 engine's headline step, the library as built against tools/ab/<name>.so, interleaved.
# usage (on the GPU box): tools/ab_engine.sh <name> [rounds=3]
set -e
cd "$(dirname "$0")/.."
F="--engine f16x3 --no-cpu-baseline --no-cascade --no-geometry --no-whole-job --no-other-engine"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["roofline"]["avg_launch_ms"],5), "batched", round(d["batched"]["value"],1), "16k", round(d["pair_16k"]["value"],1), "rescans", d["rescanned_queries_per_pair"])'
for k in $(seq 1 ${2:-3}); do
  timeout -k 10 200 python3 bench.py $F | python3 -c "$P" "lib   "
  SFM_HIP_LIBRARY=tools/ab/$1.so timeout -k 10 200 python3 bench.py $F | python3 -c "$P" "$1"
done
