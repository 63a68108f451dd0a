#!/bin/bash
# Headline, batched and 16k legs of the bench, the library as built against tools/ab/<name>.so, interleaved.
# usage (on the GPU box): tools/ab_legs.sh <name> [rounds=3]
set -e
cd "$(dirname "$0")/.."
F="--no-cpu-baseline --no-cascade --no-geometry --no-whole-job --no-other-engine"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), round(d["roofline"]["frac"],4), "batched", round(d["batched"]["value"],1), "16k", round(d["pair_16k"]["value"],1))'
for k in $(seq 1 ${2:-3}); do
  timeout -k 10 200 python3 bench.py $F | python3 -c "$P" "lib   "
  SFM_HIP_LIBRARY=tools/ab/$1.so timeout -k 10 200 python3 bench.py $F | python3 -c "$P" "$1"
done
