#!/bin/bash
# round 6: x6gemm_kernel variants -> gpurun_out/r6/x6_ab2.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_ab2.txt; : > $OUT
timeout 900 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu -k "six_piece or wide_layers or x6" 2>&1 | tail -3 >> $OUT
for e in "BSVI_X6_VAR=0" "BSVI_X6_VAR=1" "BSVI_X6_VAR=2" "BSVI_X6_VAR=3" "BSVI_X6_VAR=4" "BSVI_X6_DEBUG=4" "BSVI_X6_DEBUG=5" $X6_EXTRA; do
  echo "== $e single products" >> $OUT
  env $e timeout 300 python3 tools/r6/x6_probe.py 2>&1 | grep "^M" >> $OUT
done
for cfg in "BSVI_X6_VAR=0" "BSVI_X6_VAR=${X6_BEST:-1}"; do
  echo "== cfg5 $cfg BSVI_X6_MODES=3" >> $OUT
  env $cfg BSVI_X6_MODES=3 timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $OUT 2>&1
done
cat $OUT
