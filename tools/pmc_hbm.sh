# HBM traffic per dispatch (MI355X_MICROARCH.md "HBM" section: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes,
# kernel-trace only).  usage: bash tools/pmc_hbm.sh <tag> <bench.py args...>   -> gpurun_out/r2/<tag>_hbm.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
rows = []
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_${tag}_%s/**/*counter_collection.csv" % ctr, recursive=True)[0]
    acc = collections.defaultdict(list)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, d), v in per.items():
        acc[k].append(v)
    for k, v in acc.items():
        if "bsvi" in k:
            rows.append((k, ctr, len(v), sum(v) / len(v), max(v), sum(v)))
with open("$OUT/${tag}_hbm.csv", "w") as o:
    o.write("kernel,counter,dispatches,mean_KB_per_dispatch,max_KB,total_KB\n")
    for k, c, n, m, mx, tot in rows:
        o.write('"%s",%s,%d,%.4f,%.4f,%.4f\n' % (k, c, n, m, mx, tot))
print(open("$OUT/${tag}_hbm.csv").read())
PY
