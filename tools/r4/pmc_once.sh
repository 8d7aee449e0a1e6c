#!/bin/bash
# round 4: counters per launch of a few cfg 4 iterations.  usage: bash tools/r4/pmc_once.sh <tag> COUNTER...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-pmc}; shift
OUT=$ROOT/gpurun_out/r4/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc -- python3 $ROOT/tools/r4/cfg4_once.py 4 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc.txt", "a") as w:
    for k, d in acc.items():
        if "dense" in k or "xgemm" in k:
            line = "%-30s " % k + " ".join("%s=%.0f" % (c, sum(v) / len(v)) for c, v in sorted(d.items()))
            print(line); w.write(line + "\n")
PY
rm -rf $OUT/pmc
