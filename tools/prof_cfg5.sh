# kernel-time profile of the amortised (cfg5) iteration; usage: bash tools/prof_cfg5.sh <tag> [extra bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2/$TAG/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/$TAG/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r2/$TAG/prof -name "*kernel_stats.csv" | head -1 | xargs head -16 | cut -c1-150
python3 - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r2/$TAG/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# one iteration's launches in order (the last iteration of the run)
names = [r["Kernel_Name"] for r in rows]
last = len(names) - 1 - names[::-1].index([n for n in names if "amort_rows" in n][0])
for r in rows[last:]:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%8.1f us  grid %-18s %s" % (dur, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"], r["Kernel_Name"][:70]))
PY
