#!/bin/bash
# round 6: the amortised path after the x6gemm_kernel rebuild: its GPU tests, cfg 5's line (with live traffic), kernel stats, SQ counters, timeline
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6/cfg5; mkdir -p $OUT; cd $ROOT
timeout 1500 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu 2>&1 | tail -5 > $OUT/tests.txt
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "importance" 2>&1 | tail -5 >> $OUT/tests.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 >> $OUT/tests.txt
python3 bench.py --workload cfg5 --steps 100 --warmup 10 --other-configs off > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o run -- python3 $ROOT/bench.py --workload cfg5 --steps 60 --warmup 5 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
cp $(find /tmp/prof5 -name "*kernel_stats.csv" | head -1) $OUT/cfg5_kernel_stats.csv
rm -rf /tmp/pmc5
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/pmc5 -- python3 $ROOT/bench.py --workload cfg5 --steps 20 --warmup 3 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
files = glob.glob("/tmp/pmc5/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
with open("$OUT/cfg5_pmc_sq.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches\n")
    for k, dd in acc.items():
        if "bsvi" in k:
            for c, v in sorted(dd.items()):
                o.write('"%s",%s,%.1f,%d\n' % (k[:90], c, sum(v) / len(v), len(v)))
PY
cd $ROOT
sed -i 's#gpurun_out/r5#gpurun_out/r6/cfg5#g' /dev/null
bash tools/r5/cfg5_timeline.sh > /dev/null 2>&1; cp gpurun_out/r5/cfg5_timeline.txt $OUT/cfg5_timeline.txt 2>/dev/null
cat $OUT/tests.txt; python3 -c "
import json; d=json.loads(open('$OUT/bench_cfg5.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
grep "x6gemm\|xgemm\|x6tn" $OUT/cfg5_pmc_sq.csv | grep "MFMA_BUSY\|SQ_BUSY\|BANK" | cut -c1-150
cat $OUT/cfg5_timeline.txt
