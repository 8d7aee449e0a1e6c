#!/bin/bash
# round 4: where the time of dense_xfwd / dense_xbwd goes — the profiling variants (BSVI_XF_DEBUG / BSVI_XB_DEBUG) under the kernel trace.
# usage: bash tools/r4/dense_variants.sh <tag> "<f-modes>" "<b-modes>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/${1:-variants}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in ${2:-0 1 2 3}; do for b in ${3:-0}; do
  export BSVI_XF_DEBUG=$f BSVI_XB_DEBUG=$b
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$f$b -- python3 $ROOT/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --traffic off > /dev/null 2>&1
  echo "== xf=$f xb=$b" >> $OUT/variants.txt
  python3 - "$(find $OUT/prof_$f$b -name '*kernel_stats.csv' | head -1)" >> $OUT/variants.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_" in r["Name"]: print("  %-28s calls %5s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/prof_$f$b
done; done
cat $OUT/variants.txt
