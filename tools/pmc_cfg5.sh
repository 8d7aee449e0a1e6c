# SQ counters of the amortised (cfg5) GEMM kernels; usage: bash tools/pmc_cfg5.sh <tag>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/$1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/cfg5_pmc_sq.csv", "w") as o:
    o.write("kernel,counter,dispatches,mean\n")
    for k, d in acc.items():
        if "bsvi" in k:
            for c, v in d.items():
                o.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
print(open("$OUT/cfg5_pmc_sq.csv").read())
PY
