#!/bin/bash
# round 4: the driver's own command, its JSON line kept under gpurun_out/r4/ and summarised.  usage: bash tools/r4/driver_like.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like${1:+_$1}.json 2> $OUT/bench_driver_like${1:+_$1}.err
echo "wall seconds: $(( $(date +%s) - T0 ))"
python - "$OUT/bench_driver_like${1:+_$1}.json" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
print("cfg1 value %.0f it/s, %.3f us wall/step, %.3f us device/step; cold" % (l["value"], l["ms_per_step"] * 1e3, l["device_ms_per_step"] * 1e3), l.get("cold_start"))
print("roofline", {k: v for k, v in l["roofline"].items() if k in ("achieved", "frac", "traffic", "launch_ms")})
print("issue", l["roofline"].get("issue"))
print("probe s", l["roofline"].get("traffic_probe_seconds"), l["roofline"].get("traffic_note"))
for k, v in l.get("other_configs", {}).items():
    r = v.get("roofline", {})
    print(k, "ms/step %s device %s frac %s traffic %s cold %s %s" % (v.get("ms_per_step"), v.get("device_ms_per_step"), r.get("frac"), r.get("traffic"),
          v.get("cold_start", {}).get("ms_per_step"), v.get("error", "")), "issue", (r.get("issue") or {}).get("frac"))
print("cpu", l.get("cpu_baseline", {}).get("value"))
PY
