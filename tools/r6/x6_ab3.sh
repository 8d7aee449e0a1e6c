#!/bin/bash
# round 6: x6gemm_kernel, compiler-ordered loop (BSVI_X6_VAR=0) against the hand-ordered loop (1) -> gpurun_out/r6/x6_ab3.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_ab3.txt; : > $OUT
for v in 0 1; do
  echo "== BSVI_X6_VAR=$v tests" >> $OUT
  BSVI_X6_VAR=$v timeout 900 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu -k "six_piece or wide_layers or x6" 2>&1 | tail -3 >> $OUT
  echo "== BSVI_X6_VAR=$v single products" >> $OUT
  BSVI_X6_VAR=$v timeout 300 python3 tools/r6/x6_probe.py 2>&1 | grep "^M" >> $OUT
  echo "== cfg5 BSVI_X6_VAR=$v BSVI_X6_MODES=3" >> $OUT
  BSVI_X6_VAR=$v BSVI_X6_MODES=3 timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $OUT 2>&1
done
cat $OUT
