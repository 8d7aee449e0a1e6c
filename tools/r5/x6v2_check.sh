#!/bin/bash
# round 5: x6gemm2_kernel (staging waves + product waves) against x6gemm_kernel: the single products at cfg 5's shapes, the
# accuracy / determinism tests of the amortised path, cfg 5 itself.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for v in 0 1; do echo "BSVI_X6_V2=$v"; BSVI_X6_V2=$v python3 tools/r4/x6_probe.py 2>&1 | grep "^M" | cut -c1-150; done
BSVI_X6_V2=1 timeout 900 python -m pytest tests/test_gpu_amortized.py -m gpu -q -x -k "six_piece or x6 or exact or determin or golden or parity" 2>&1 | tail -5
for v in 0 1; do
  BSVI_X6_V2=$v timeout 300 python bench.py --workload cfg5 --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('BSVI_X6_V2=$v cfg5 %.1f us per iteration' % (l['device_ms_per_step']*1e3))"
done
