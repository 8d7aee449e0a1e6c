#!/bin/bash
# kernel-time breakdown of cfg 5
cd /root/repo
mkdir -p gpurun_out/prof5
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -o p5 -- python3 bench.py --workload cfg5 --steps 50 --warmup 10 --spinup-ms 0 --no-cpu-baseline --traffic off > gpurun_out/prof5/bench.log 2>&1
f=$(find gpurun_out/prof5 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 gpurun_out/prof5/bench.log | cut -c1-200
