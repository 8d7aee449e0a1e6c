#!/bin/bash
# round 5: the whole GPU suite, then the driver's bench command.  usage: bash tools/r5/suite_and_bench.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/${1:-full}
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench.err
tail -15 $OUT/suite.txt
python - "$OUT/bench_driver_like.json" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg1: value %.0f it/s, %.3f us/step wall, %.3f device" % (l["value"], l["ms_per_step"] * 1e3, l["device_ms_per_step"] * 1e3))
for k, v in l.get("other_configs", {}).items():
    print("%s: %.2f us/step device, roofline frac %.4f" % (k, v["device_ms_per_step"] * 1e3, v["roofline"]["frac"]))
PY
