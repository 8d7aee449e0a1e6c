# PMC counters of the GEMM probe launches; usage: bash tools/pmc_probe.sh <tag> "<counter list>"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/$1
rm -rf $OUT/pmc
rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "gemm_kernel" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("gemm_kernel")[1][:12], r["Grid_Size"])
    acc.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
