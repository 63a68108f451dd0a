#!/bin/bash
# Runs on the GPU box: bench line, rocprofv3 kernel statistics of the same command, and the
# hardware-counter passes.  Results land under gpurun_out/; copy what is to be kept into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/collect; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -u $R/bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300 || exit 1
timeout -k 10 600 python -u $R/bench.py --no-cpu-baseline --no-other-engine --no-whole-job --no-cascade --no-geometry --untimed-pass 2>>$OUT/bench.err > $OUT/bench_untimed_pass.json || exit 1
timeout -k 10 600 python -u $R/bench.py --no-cpu-baseline --no-other-engine --no-whole-job --no-cascade --no-geometry --untimed-pass --engine f16x3 2>>$OUT/bench.err > $OUT/bench_untimed_pass_f16x3.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-cascade --no-geometry --no-batched --no-whole-job > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv && cat $OUT/kernel_stats.csv
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-profile-events --no-cascade --no-geometry --no-batched --no-whole-job"
run() { local name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $R/bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/pmc_$name.log; return 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
run b GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES &&
run c FETCH_SIZE && run d WRITE_SIZE &&
python3 $R/tools/pmc_summary.py $OUT/pmc $OUT/traffic.json > $OUT/pmc_summary.txt && grep -A22 "k_distance_topk" $OUT/pmc_summary.txt
