#!/bin/bash
# round 4: where the time of bsvi_mvn_kernel goes — the kernel cut off behind step k (BSVI_SPEC_DEFINES="#define MVN_STOP_AFTER k"),
# kernel durations from the trace.  usage: bash tools/r4/mvn_steps.sh "<dims>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/mvn_steps
mkdir -p $OUT
cat > /tmp/mvn_one.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import torch
from brancher_amd import engine, workloads as W
D = int(sys.argv[1])
c = engine.compile_model(W.build_gp_hyperparameters(W.native_api(), n=D, jitter=5e-2), None, "pathwise")
for _ in range(6):
    c.evaluate(512, seed=1)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp BSVI_JIT_CACHE=0
for D in ${1:-32 128}; do
  for k in 1 2 3 4 5 0; do
    if [ $k = 0 ]; then unset BSVI_SPEC_DEFINES; else export BSVI_SPEC_DEFINES="#define MVN_STOP_AFTER $k"; fi
    rm -rf /tmp/prof_mvn
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mvn -o run -- python3 /tmp/mvn_one.py $D > /dev/null 2>&1
    f=$(find /tmp/prof_mvn -name "*kernel_stats.csv" | head -1)
    python3 - "$f" $D $k <<'PY' | tee -a $OUT/steps.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("bsvi_mvn_kernel"):
        print("D = %s, through step %s: %.1f us (min %.1f)" % (sys.argv[2], sys.argv[3] if sys.argv[3] != "0" else "6 (whole)", float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
