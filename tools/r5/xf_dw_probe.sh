#!/bin/bash
# round 5: dense_xfwd with eight drawing waves (two per SIMD, 768 threads) against four — cfg 4 per iteration, and the dense tests with it.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/xf_dw
mkdir -p $OUT
cd $ROOT
for dw in 4 8 4 8; do
  BSVI_XF_DW=$dw python bench.py --workload cfg4 --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('BSVI_XF_DW=$dw: %.1f us/step, final loss %r' % (l['ms_per_step']*1e3, l['final_loss']))" | tee -a $OUT/cfg4.txt
done
BSVI_XF_DW=8 timeout 900 python -m pytest tests/test_gpu_dense_fused.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/tests_dw8.txt
