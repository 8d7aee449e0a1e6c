# PMC pass on the dense (cfg4) iteration; usage: bash tools/pmc_cfg4.sh <tag>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/$1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "dense_forward" in k or "dense_backward" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
