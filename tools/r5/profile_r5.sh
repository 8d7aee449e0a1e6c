#!/bin/bash
# Round-5 evidence -> gpurun_out/r5/evidence (copied into profiles/r5/): the bench as the driver runs it (headline + cfg 2-5 in one
# line), the long default run, cfg 4 (Pathwise / BlackBox) and cfg 5; rocprofv3 kernel stats of the same commands; SQ counters
# (MFMA-busy, MOPS, LDS conflicts) of the cfg 4 / cfg 5 launches; L2 counters of cfg 4; the Bayesian neural network at the example's size.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/evidence
mkdir -p $OUT
cd $ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
python3 bench.py --other-configs off > $OUT/bench_default.json 2>/dev/null
for w in cfg4 cfg5; do
  python3 bench.py --workload $w --steps 100 --warmup 10 --other-configs off > $OUT/bench_$w.json 2>/dev/null
done
python3 bench.py --workload cfg4 --estimator blackbox --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_cfg4_blackbox.json 2>/dev/null
python3 tools/r5/bnn_timing.py > $OUT/bnn_timing.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
run_stats () {   # tag, bench args
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
  cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv
  rm -rf /tmp/prof_$tag
  head -5 $OUT/${tag}_kernel_stats.csv | cut -c1-150
}
run_stats default
run_stats driver_like --steps 20 --warmup 5
run_stats cfg4 --workload cfg4 --steps 100 --warmup 10 --spinup-ms 0
run_stats cfg5 --workload cfg5 --steps 60 --warmup 5 --spinup-ms 0
for w in cfg4 cfg5; do
  rm -rf /tmp/pmc_sq_$w
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sq_$w -- python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
files = glob.glob("/tmp/pmc_sq_$w/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
with open("$OUT/${w}_pmc_sq.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches\n")
    for k, dd in acc.items():
        if "bsvi" in k or "dense" in k:
            for c, v in sorted(dd.items()):
                o.write('"%s",%s,%.1f,%d\n' % (k[:90], c, sum(v) / len(v), len(v)))
PY
  rm -rf /tmp/pmc_sq_$w
  grep "xfwd\|xbwd\|xgemm\|gemm_kernel" $OUT/${w}_pmc_sq.csv | grep "MFMA_BUSY\|SQ_BUSY\|MOPS_BF16" | cut -c1-140
done
rm -rf /tmp/pmc_l2
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pmc_l2 -- python3 $ROOT/tools/r4/cfg4_once.py 4 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
fs = glob.glob("/tmp/pmc_l2/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/cfg4_l2_counters.txt", "w") as w:
    for k, d in acc.items():
        if "dense" in k:
            w.write("%-30s " % k + " ".join("%s=%.0f" % (c, sum(v) / len(v)) for c, v in sorted(d.items())) + "\n")
PY
rm -rf /tmp/pmc_l2
cat $OUT/cfg4_l2_counters.txt $OUT/bnn_timing.txt
python3 - "$OUT/bench_driver_like.json" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg1 %.0f it/s, %.2f us wall/step, %.2f device; cold %.2f us/step" % (l["value"], l["ms_per_step"] * 1e3, l["device_ms_per_step"] * 1e3, l["cold_start"]["ms_per_step"] * 1e3))
print("issue_frac", l["roofline"].get("issue_frac"), "traffic", l["roofline"].get("traffic"))
for k, v in l.get("other_configs", {}).items():
    r = v.get("roofline", {})
    print(k, "us/step %.2f frac %s traffic %s %s" % (v.get("ms_per_step", 0) * 1e3, r.get("frac"), r.get("traffic"), v.get("error", "")))
PY
