#!/bin/bash
# round 5: dense_epilogue folded into the parameter kernel — the dense tests, then config 4 with and without the fold.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/fold
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_dense_fused.py tests/test_gpu_parity.py tests/test_gpu_c_abi.py tests/test_gpu_two_ranks.py -m gpu -q -x 2>&1 | tail -8 > $OUT/tests.txt
cat $OUT/tests.txt
for f in 1 0; do
  BSVI_DENSE_FOLD=$f timeout 300 python bench.py --workload cfg4 --steps 2000 --warmup 50 --other-configs off --no-cpu-baseline --traffic off 2> /dev/null | tail -1 > $OUT/cfg4_fold$f.json
  python - $OUT/cfg4_fold$f.json $f <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read())
print("fold", sys.argv[2], "cfg4 %.2f us/step device, %.2f wall" % (l["device_ms_per_step"] * 1e3, l["ms_per_step"] * 1e3))
PY
done
