#!/bin/bash
# round 5: what a 20-iteration call of cfg 1 consists of on the device — durations of the bsvi_spec_kernel dispatches of
# `bench.py --steps 20 --warmup 5` (the last 20-iteration launches are the cold call and the timed one; the long ones are the spin-up)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c1trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c1trace -o run -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --other-configs off --traffic off > $OUT/cfg1_call_trace_bench.json 2>/dev/null
python3 - <<'PY' > $OUT/cfg1_call_trace.txt
import csv, glob
f = glob.glob("/tmp/c1trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "bsvi_spec_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("bsvi_spec_kernel dispatches:", len(d))
print("first five (warm-up 5 iterations, cold 20):", " ".join("%.1f" % x for x in d[:5]))
short = [x for x in d if x < 400]
print("20-iteration launches (us):", " ".join("%.1f" % x for x in short[-8:]))
long_ = [x for x in d if x >= 400]
if long_:
    print("spin-up launches: %d, mean %.1f us" % (len(long_), sum(long_) / len(long_)))
PY
cat $OUT/cfg1_call_trace.txt
python3 -c "
import json
l=json.loads(open('$OUT/cfg1_call_trace_bench.json').read().strip().splitlines()[-1])
print('bench under the profiler: %.2f us wall per step, %.2f device' % (l['ms_per_step']*1e3, l['device_ms_per_step']*1e3))"
rm -rf /tmp/c1trace
