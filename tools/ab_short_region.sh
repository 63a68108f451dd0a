#!/bin/bash
# The driver's form of the bench (20 steps, 5 warm-up), headline leg only, several times, with and
# without per-launch kernel timing: what the short region's edges and the timing markers cost.
set -e
cd "$(dirname "$0")/.."
F="--steps 20 --warmup 5 --no-cpu-baseline --no-other-engine --no-whole-job --no-batched --no-cascade --no-geometry"
for k in 1 2 3 4; do
  timeout -k 10 120 python3 bench.py $F | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('timed   ', round(d['value'],1), round(d['ms_per_step']*1e3,2), d['roofline'] and round(d['roofline']['frac'],4))"
  timeout -k 10 120 python3 bench.py $F --no-profile-events | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('untimed ', round(d['value'],1), round(d['ms_per_step']*1e3,2))"
done
