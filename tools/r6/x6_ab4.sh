#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_ab4.txt; : > $OUT
for st in 0 64 128 192 256; do
  echo "== BSVI_X6_VAR=1 BSVI_X6_STAGGER=$st single products" >> $OUT
  BSVI_X6_VAR=1 BSVI_X6_STAGGER=$st timeout 300 python3 tools/r6/x6_probe.py 2>&1 | grep "^M" >> $OUT
done
BSVI_X6_STAGGER=128 BSVI_X6_DEBUG=9 python3 tools/r6/x6_stamps.py 512 256 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
