# instruction mix of one fused ELBO launch (1 wave: number_samples=64), stepwise mode; usage: bash tools/pmc_insts.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_insts -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode stepwise --samples 64 --steps 200 --warmup 20 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_insts/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_sq_insts_N64.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches\n")
    for k, d in acc.items():
        if "bsvi" in k:
            for c, v in sorted(d.items()):
                o.write('"%s",%s,%.1f,%d\n' % (k, c, sum(v) / len(v), len(v)))
print(open("$OUT/pmc_sq_insts_N64.csv").read())
PY
