#!/bin/bash
# Round-6 evidence -> gpurun_out/r6/evidence (copied into profiles/r6/): the whole GPU suite + smoke(), the bench as the driver runs it
# (headline + cfg 2-5 + the two BlackBox lines + the Bayesian neural network in one line), the long default run, cfg 4 and cfg 5 alone
# (live traffic), rocprofv3 kernel stats of the same commands, SQ counters (MFMA-busy, MOPS, LDS conflicts) of the cfg 4 / cfg 5
# launches, the kernel timeline of one cfg-5 iteration, README's perform_inference call timed cold / cached.
# usage: bash tools/r6/profile_r6.sh [nosuite]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6/evidence
mkdir -p $OUT
cd $ROOT
if [ "$1" != "nosuite" ]; then
  timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $OUT/gpu_suite.txt
  timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> $OUT/gpu_suite.txt
  cat $OUT/gpu_suite.txt
fi
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
python3 bench.py --other-configs off > $OUT/bench_default.json 2>/dev/null
for w in cfg4 cfg5; do
  python3 bench.py --workload $w --steps 100 --warmup 10 --other-configs off > $OUT/bench_$w.json 2>/dev/null
  python3 bench.py --workload $w --estimator blackbox --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_${w}_blackbox.json 2>/dev/null
done
timeout 600 python3 tools/r6/perform_inference_readme.py > $OUT/perform_inference_readme.txt 2>&1
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
run_stats bnn_cfg4scale --workload bnn_cfg4scale --steps 60 --warmup 5 --spinup-ms 0
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
                o.write('"%s",%s,%.1f,%d\n' % (k[:110], c, sum(v) / len(v), len(v)))
PY
  rm -rf /tmp/pmc_sq_$w
  grep "xfwd\|xbwd\|x6gemm\|x6tn" $OUT/${w}_pmc_sq.csv | grep "MFMA_BUSY\|SQ_BUSY\|BANK" | cut -c1-170
done
cd $ROOT
bash tools/r5/cfg5_timeline.sh > /dev/null 2>&1; cp gpurun_out/r5/cfg5_timeline.txt $OUT/cfg5_timeline.txt 2>/dev/null
cat $OUT/perform_inference_readme.txt
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
l = json.loads(open(out + "/bench_driver_like.json").read().strip().splitlines()[-1])
print("cfg1 %.0f it/s, %.2f us wall/step, %.2f device; cold %.2f us/step" % (l["value"], l["ms_per_step"] * 1e3, l["device_ms_per_step"] * 1e3, l["cold_start"]["ms_per_step"] * 1e3))
print("issue_frac", l["roofline"].get("issue_frac"), "traffic", l["roofline"].get("traffic"), "cpu", json.dumps(l.get("cpu_baseline"))[:400])
for k, v in l.get("other_configs", {}).items():
    r = v.get("roofline", {})
    print(k, "us/step %.2f frac %s traffic %s %s" % (v.get("ms_per_step", 0) * 1e3, r.get("frac"), r.get("traffic"), v.get("error", "")))
for w in ("cfg4", "cfg5", "cfg4_blackbox", "cfg5_blackbox"):
    try:
        d = json.loads(open(out + "/bench_%s.json" % w).read().strip().splitlines()[-1])
        print(w, "alone: us/step %.2f frac %.4f traffic %s" % (d["ms_per_step"] * 1e3, d["roofline"]["frac"], d["roofline"].get("traffic")))
    except Exception as e:
        print(w, "no line:", e)
PY
