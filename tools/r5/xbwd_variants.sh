#!/bin/bash
# round 5: dense_xbwd under the kernel trace for a list of "VAR=value" settings.  usage: bash tools/r5/xbwd_variants.sh <tag> "<setting> <setting> ..."
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/${1:-xbwd_variants}
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_dense_fused.py -x -q 2>&1 | tail -5 > $OUT/tests_fused.txt; cat $OUT/tests_fused.txt; fi
cd /tmp && export TMPDIR=/tmp
i=0
for setting in ${2:-BSVI_XB_STAGGER=0}; do
  i=$((i+1))
  export $setting
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$i -- python3 $ROOT/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --traffic off > /dev/null 2>&1
  echo "== $setting" >> $OUT/variants.txt
  python3 - "$(find $OUT/prof_$i -name '*kernel_stats.csv' | head -1)" >> $OUT/variants.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dense_" in r["Name"]: print("  %-28s calls %5s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/prof_$i
done
cat $OUT/variants.txt
