#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bnn or refused" 2>&1 | tail -15 > $OUT/check3.txt
timeout 1500 python3 -m pytest tests/test_gpu_bnn.py tests/test_gpu_specialised.py -x -q -m gpu 2>&1 | tail -8 >> $OUT/check3.txt
cat $OUT/check3.txt
