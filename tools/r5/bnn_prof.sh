#!/bin/bash
# round 5: kernel times of the Bayesian-neural-network iteration (tools/r5/bnn_timing.py under the kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/bnn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/r5/bnn_timing.py > $OUT/timing.txt 2>/dev/null
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bnn_kernel_stats.csv; rm -rf $OUT/prof
python3 - $OUT/bnn_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-60s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cat $OUT/timing.txt
