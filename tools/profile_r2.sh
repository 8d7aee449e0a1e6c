#!/bin/bash
# Round-2 evidence for the default bench (BASELINE config 1, specialised kernel, in-kernel loop) and its launch-per-
# iteration mode: rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and the SQ
# instruction mix.  Everything lands in gpurun_out/r2/; copy what is cited into profiles/r2/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_stats () {   # tag, bench args
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $OUT/${tag}_prof.log 2>&1
  tail -1 $OUT/${tag}_prof.log > $OUT/bench_${tag}_under_rocprof.json
  cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv
  head -4 $OUT/${tag}_kernel_stats.csv | cut -c1-160
}
run_stats default
run_stats stepwise --mode stepwise --steps 5000
run_stats cfg3 --workload cfg3 --steps 500 --warmup 20
run_stats cfg1_big --workload cfg1_big --steps 200 --warmup 20
cd $GRAFT_REPO_ROOT
bash tools/pmc_hbm.sh cfg1_spec | tail -4
bash tools/pmc_hbm.sh cfg1_spec_stepwise --mode stepwise --steps 2000 | tail -4
# instruction mix of ONE iteration: a 64-sample launch per iteration (one wave), specialised kernel
cd /tmp
rm -rf /tmp/pmc_insts
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_insts -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode stepwise --samples 64 --steps 200 --warmup 20 --spinup-ms 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_insts/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_sq_insts_N64.csv", "w") as o:
    o.write("kernel,counter,mean_per_launch,launches\n")
    for k, d in acc.items():
        if "bsvi" in k:
            for c, v in sorted(d.items()):
                o.write('"%s",%s,%.1f,%d\n' % (k, c, sum(v) / len(v), len(v)))
print(open("$OUT/pmc_sq_insts_N64.csv").read())
PY
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2>/dev/null; cut -c1-260 $OUT/bench_default.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2>/dev/null; cut -c1-260 $OUT/bench_driver_like.json
python3 bench.py --workload cfg3 --steps 2000 > $OUT/bench_cfg3.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg3.json
python3 bench.py --workload cfg2 --steps 2000 > $OUT/bench_cfg2.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg2.json
python3 bench.py --workload cfg1_big --steps 300 --warmup 20 > $OUT/bench_cfg1_big.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg1_big.json
