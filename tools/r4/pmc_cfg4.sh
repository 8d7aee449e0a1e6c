#!/bin/bash
# round 4: SQ counters per launch of the cfg 4 iteration.  usage: bash tools/r4/pmc_cfg4.sh <tag> [counters...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-pmc4}; shift
OUT=$ROOT/gpurun_out/r4/$TAG
mkdir -p $OUT
CNT=${@:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --workload cfg4 --steps 6 --warmup 2 --no-cpu-baseline --traffic off --other-configs off --spinup-ms 0 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_sq.csv", "a") as w:
    for k, d in acc.items():
        if "dense" in k or "xgemm" in k:
            line = "%-30s " % k + " ".join("%s=%.0f" % (c, sum(v) / len(v)) for c, v in sorted(d.items()))
            print(line); w.write(line + "\n")
PY
rm -rf $OUT/pmc
