mkdir -p gpurun_out/r5/graph
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sh
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -o run -- python3 $GRAFT_REPO_ROOT/tools/r5/sharded_call_probe.py > /dev/null 2>&1
cp $(find /tmp/prof_sh -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r5/graph/sharded_kernel_stats.csv
python3 - <<'PY'
import csv, glob, os
f = glob.glob("/tmp/prof_sh/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a window of 14 consecutive dispatches from the middle of the kept-graph phase
mid = len(rows) // 6
t0 = int(rows[mid]["Start_Timestamp"])
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/graph/sharded_timeline.txt", "w") as o:
    for r in rows[mid:mid + 14]:
        o.write("%9.2f us  +%6.2f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
PY
rm -rf /tmp/prof_sh
