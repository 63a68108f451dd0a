#!/bin/bash
# first contact with the GPU: does the library load, do the parity tests pass
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu --timeout 180 --timeout-method thread 2>&1 | tee gpurun_out/first_tests.log | tail -60
