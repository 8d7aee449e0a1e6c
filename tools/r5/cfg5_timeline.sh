#!/bin/bash
# kernel timeline of one cfg 5 iteration (start offsets, durations, queue) -> gpurun_out/r5/cfg5_timeline.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --workload cfg5 --steps 40 --warmup 30 --no-cpu-baseline --other-configs off --traffic off > /tmp/tl/bench.log 2>&1
mkdir -p gpurun_out/r5
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last complete iteration: from an amort_rows launch to the next
starts = [i for i, r in enumerate(rows) if 'amort_rows' in r['Kernel_Name'] or 'amort_head' in r['Kernel_Name']]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]['Start_Timestamp'])
out = open('gpurun_out/r5/cfg5_timeline.txt', 'w')
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].replace('bsvi_amort_impl::', '')[:60]
    out.write('%8.1f %8.1f %7.1f us  q%s  %s  grid %s\n' % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name, r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
out.write('iteration %.1f us\n' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
out.close()
print(open('gpurun_out/r5/cfg5_timeline.txt').read())
PY
