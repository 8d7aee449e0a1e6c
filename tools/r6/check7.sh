#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_specialised.py -x -q -m gpu -k "prf" 2>&1 | tail -25 > $OUT/check7.txt
timeout 900 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu -k "fused_likelihood" 2>&1 | tail -5 >> $OUT/check7.txt
timeout 900 python3 -m pytest tests/test_gpu_mvn.py -x -q -m gpu 2>&1 | tail -3 >> $OUT/check7.txt
cat $OUT/check7.txt
