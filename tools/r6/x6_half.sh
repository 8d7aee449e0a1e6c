#!/bin/bash
# round 6: x6gemm_kernel with the first round's second workgroups on half tiles (BSVI_X6_HALF=0 off, 1 workgroups 256..511, 2 the odd ones)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_half.txt; : > $OUT
timeout 1500 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu 2>&1 | tail -3 >> $OUT
for e in "BSVI_X6_HALF=0" "BSVI_X6_HALF=1" "BSVI_X6_HALF=2"; do
  echo "== $e single products" >> $OUT
  env $e timeout 300 python3 tools/r6/x6_probe.py 2>/dev/null >> $OUT
done
for e in "BSVI_X6_HALF=0" "BSVI_X6_HALF=1" "BSVI_X6_HALF=2" "BSVI_X6_HALF=0" "BSVI_X6_HALF=1" "BSVI_X6_HALF=2"; do
  echo "== cfg5 $e" >> $OUT
  env $e timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])" >> $OUT 2>&1
done
cat $OUT
