#!/bin/bash
# Round-3 evidence: the bench as the driver runs it (headline + cfg 2-5 in one line, live PMC traffic), the long default run,
# rocprofv3 kernel stats of cfg 1 / 4 / 5 (exact-data paths and, for comparison, the f32-input kernels), the batched
# multivariate-normal kernel, and the SQ MFMA-busy counters of the two matrix-core paths.  Lands in gpurun_out/r3/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err
python3 bench.py --other-configs off > $OUT/bench_default.json 2>/dev/null
for w in cfg4 cfg5; do
  python3 bench.py --workload $w --steps 100 --warmup 10 --other-configs off > $OUT/bench_$w.json 2>/dev/null
done
BSVI_DENSE_XGEMM=0 python3 bench.py --workload cfg4 --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_cfg4_f32_kernels.json 2>/dev/null
BSVI_AMORT_XGEMM=0 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_cfg5_f32_kernels.json 2>/dev/null
python3 bench.py --workload cfg4 --estimator blackbox --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_cfg4_blackbox.json 2>/dev/null
python3 bench.py --workload cfg4_unit --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off > $OUT/bench_cfg4_unit.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
run_stats () {   # tag, bench args
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --other-configs off --traffic off > $OUT/${tag}_prof.log 2>&1
  cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv
  head -4 $OUT/${tag}_kernel_stats.csv | cut -c1-150
}
run_stats default
run_stats driver_like --steps 20 --warmup 5
run_stats cfg4 --workload cfg4 --steps 100 --warmup 10 --spinup-ms 0
run_stats cfg5 --workload cfg5 --steps 60 --warmup 5 --spinup-ms 0
# MFMA-busy of the matrix-core launches (SQ counters, own pass)
for w in cfg4 cfg5; do
  rm -rf /tmp/pmc_sq_$w
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_sq_$w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 3 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
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
  head -30 $OUT/${w}_pmc_sq.csv | cut -c1-140
done
# the batched multivariate-normal kernel
cd $GRAFT_REPO_ROOT
python3 - > $OUT/gp_timings.txt <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from brancher_amd import engine, workloads as W
for n_points, n in ((32, 512), (64, 512), (100, 512), (128, 512)):
    c = engine.compile_model(W.build_gp_hyperparameters(W.native_api(), n=n_points, jitter=5e-2), None, "pathwise")
    c.train(5, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.train(50, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    print("gp_hyperparameters D=%d number_samples=%d: %.1f us per iteration (base program + bsvi_mvn_kernel + full program + optimizer)" % (n_points, n, (time.perf_counter() - t0) / 50 * 1e6))
PY
cat $OUT/gp_timings.txt
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3/bench_driver_like.json") if l.startswith("{")][-1])
print("driver-like", round(d["value"]), d["ms_per_step"], d["device_ms_per_step"], d["roofline"]["traffic"])
for k, v in d.get("other_configs", {}).items():
    print(k, v.get("error") or (round(v["ms_per_step"]*1e3, 1), v["roofline"]["frac"], v["roofline"]["traffic"]))
for f in ("bench_default", "bench_cfg4", "bench_cfg5", "bench_cfg4_f32_kernels", "bench_cfg5_f32_kernels", "bench_cfg4_blackbox", "bench_cfg4_unit"):
    try:
        d = json.loads([l for l in open("gpurun_out/r3/%s.json" % f) if l.startswith("{")][-1])
        print(f, round(d["value"]), round(d["ms_per_step"]*1e3, 2), "us", d["roofline"]["frac"])
    except Exception as e:
        print(f, "missing", e)
PY
