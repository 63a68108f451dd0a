set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/k20; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do timeout -k 10 200 python -u $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-engine --no-whole-job --no-cascade --no-batched 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 value', d['value'], 'ms/step', d['ms_per_step'])" || exit 1; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-engine --no-whole-job --no-cascade --no-batched > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 $R/tools/region_timeline.py $OUT/trace/*/*_kernel_trace.csv 20 | tee $OUT/timeline.txt | head -20; tail -12 $OUT/timeline.txt
