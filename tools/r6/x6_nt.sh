#!/bin/bash
# round 6: the C stores of x6gemm_kernel as non-temporal stores (BSVI_X6_DEBUG=6) — single products, then cfg 5
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_nt.txt; : > $OUT
for e in "BSVI_X6_DEBUG=0" "BSVI_X6_DEBUG=6" "BSVI_X6_DEBUG=0" "BSVI_X6_DEBUG=6"; do
  echo "== $e single products" >> $OUT
  env $e timeout 300 python3 tools/r6/x6_probe.py 2>/dev/null >> $OUT
done
for e in "BSVI_X6_DEBUG=0" "BSVI_X6_DEBUG=6" "BSVI_X6_DEBUG=0" "BSVI_X6_DEBUG=6"; do
  echo "== cfg5 $e" >> $OUT
  env $e timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])" >> $OUT 2>&1
done
cat $OUT
