#!/bin/bash
# round 5: dense_xbwd against the number of column groups (BSVI_XB_GROUPS): 36 groups x 7 feature tiles put 35 workgroups on the
# 32 CUs of XCDs 0-3 (workgroup i runs on XCD i % 8) — a second round.  usage: bash tools/r5/xbwd_groups.sh <tag> "<groups...>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/${1:-xbwd_groups}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in ${2:-0 32 24 16}; do
  export BSVI_XB_GROUPS=$g
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$g -- python3 $ROOT/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --traffic off > /dev/null 2>&1
  echo "== groups=$g" >> $OUT/variants.txt
  python3 - "$(find $OUT/prof_$g -name '*kernel_stats.csv' | head -1)" >> $OUT/variants.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_" in r["Name"]: print("  %-28s calls %5s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/prof_$g
done
cat $OUT/variants.txt
