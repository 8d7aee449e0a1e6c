#!/bin/bash
# round 4: the batched multivariate-normal kernel under the profiler — kernel trace + SQ counters of a few training iterations of
# the Gaussian-process model with inferred hyper-parameters at D = 32 / 64 / 100 / 128, 512 samples.  usage: bash tools/r4/prof_mvn.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/${1:-mvn}
mkdir -p $OUT
cat > /tmp/mvn_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import torch
from brancher_amd import engine, workloads as W
api = W.native_api()
for D in (32, 64, 100, 128):
    c = engine.compile_model(W.build_gp_hyperparameters(api, n=D, jitter=5e-2), None, "pathwise")
    c.train(3, 512, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); c.train(20, 512, "Adam", lr=1e-2, seed=1); ev1.record(); torch.cuda.synchronize()
    print("D = %3d: %.1f us per training iteration at 512 samples" % (D, ev0.elapsed_time(ev1) * 1e3 / 20))
PY
cd /tmp && export TMPDIR=/tmp
python3 /tmp/mvn_run.py > $OUT/gp_timings.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 /tmp/mvn_run.py > /dev/null 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/mvn_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/pmc -- python3 /tmp/mvn_run.py > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])) if f else []:
    if "mvn" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/mvn_pmc_sq.csv", "w") as w:
    w.write("kernel,grid,lds_block,counter,mean_per_launch,launches\n")
    for (k, g, l), d in acc.items():
        for c, v in sorted(d.items()):
            w.write("%s,%s,%s,%s,%.1f,%d\n" % (k, g, l, c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/prof $OUT/pmc
cat $OUT/gp_timings.txt; head -6 $OUT/mvn_kernel_stats.csv | cut -c1-140; head -12 $OUT/mvn_pmc_sq.csv
