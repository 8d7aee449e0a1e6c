#!/bin/bash
# start / end of every kernel of ONE cfg-5 iteration, per stream (rocprofv3 kernel trace)
cd /root/repo
bash tools/r3_prof5.sh > /dev/null 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof5/p5_kernel_trace.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows)
starts = [i for i,(s,e,n,q) in enumerate(ev) if "amort_rows" in n]
it = ev[starts[-2]:starts[-1]]
t0 = it[0][0]
for s,e,n,q in it:
    print("%8.1f %8.1f  %6.1f  q%s  %s" % ((s-t0)/1e3, (e-t0)/1e3, (e-s)/1e3, q, n.replace("bsvi_amort_impl::","").replace("void ","")[:60]))
print("next iteration starts at %.1f" % ((ev[starts[-1]][0]-t0)/1e3))
PY
