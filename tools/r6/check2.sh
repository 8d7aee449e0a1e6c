#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_collective.py tests/test_gpu_c_abi.py -x -q -m gpu 2>&1 | tail -5 > $OUT/check2.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refused or kept_graph or sharded" 2>&1 | tail -5 >> $OUT/check2.txt
timeout 1500 python3 -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu 2>&1 | tail -8 >> $OUT/check2.txt
timeout 600 python3 tools/r6/perform_inference_readme.py > $OUT/perform_inference_readme.txt 2>&1
cat $OUT/check2.txt $OUT/perform_inference_readme.txt
