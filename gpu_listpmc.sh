#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt 2>&1
echo rc=$?
grep -c . $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt
