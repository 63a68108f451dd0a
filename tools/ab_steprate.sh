#!/bin/bash
# A/B of two library builds (tools/ab/<a>.so, tools/ab/<b>.so) on the single-pair step, interleaved
for i in 1 2 3; do for v in "$@"; do echo -n "$v: "; SFM_HIP_LIBRARY=tools/ab/$v.so timeout -k 10 100 python tools/steprate.py 2>/dev/null || exit 1; done; done
