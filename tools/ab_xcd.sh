#!/bin/bash
# single-pair step under each way of cutting the pair's blocks among the eight XCDs (diagnostic build xcdenv)
for i in 1 2 3; do for g in 1 2 4 8; do echo -n "row cuts $g: "; SFM_XCD_ROW_CUTS=$g SFM_HIP_LIBRARY=tools/ab/xcdenv.so timeout -k 10 100 python tools/steprate.py 2>/dev/null || exit 1; done; done
