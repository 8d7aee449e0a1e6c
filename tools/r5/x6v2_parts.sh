#!/bin/bash
# round 5: x6gemm2_kernel with one side compiled out (timing only)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for d in ${PARTS:-0 1 2 3}; do echo "BSVI_X6_V2=1 BSVI_X6_V2_DEBUG=$d  (1: no products, 2: no staging)"; BSVI_X6_V2=1 BSVI_X6_V2_DEBUG=$d python3 tools/r4/x6_probe.py 2>&1 | grep "^M" | cut -c1-95; done
