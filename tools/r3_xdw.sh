#!/bin/bash
# exact-data weight gradient: tests, then cfg 5 with and without it
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_amortized.py -x -q 2>&1 | tail -15 > gpurun_out/xdw_tests.log
for v in 1 0; do
  BSVI_AMORT_XDW=$v timeout 300 python bench.py --workload cfg5 --steps 50 --warmup 10 --no-cpu-baseline --traffic off 2>&1 | tail -1 > gpurun_out/xdw_cfg5_$v.json
done
cat gpurun_out/xdw_tests.log
python - <<'PY'
import json
for v in (1, 0):
    try:
        d = json.loads(open("gpurun_out/xdw_cfg5_%d.json" % v).read())
        print("XDW=%d" % v, d.get("ms_per_step"), d.get("value"))
    except Exception as e:
        print("XDW=%d" % v, "unreadable", e)
PY
