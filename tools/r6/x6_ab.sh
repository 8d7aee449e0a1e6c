#!/bin/bash
# round 6: x6gemm_kernel rebuilt (LDS-DMA staging, register split, 16-byte stores) against round 5's -> gpurun_out/r6/x6_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/x6_ab.txt; : > $OUT
timeout 900 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu -k "six_piece or wide_layers or x6" 2>&1 | tail -5 >> $OUT
for v in 5 6; do
  echo "== BSVI_X6_V=$v single products" >> $OUT
  BSVI_X6_V=$v timeout 300 python3 tools/r6/x6_probe.py >> $OUT 2>&1
done
for dbg in 2 3; do
  echo "== v6 BSVI_X6_DEBUG=$dbg (2: no stores, 3: no MFMAs)" >> $OUT
  BSVI_X6_DEBUG=$dbg timeout 300 python3 tools/r6/x6_probe.py >> $OUT 2>&1
done
for cfg in "BSVI_X6_V=5" "BSVI_X6_V=6" "BSVI_X6_V=6 BSVI_X6_MODES=3"; do
  echo "== cfg5 $cfg" >> $OUT
  env $cfg timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $OUT 2>&1
done
cat $OUT
