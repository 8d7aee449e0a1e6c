# SQ counters of the default run's in-kernel training loop (one launch = K iterations of BASELINE config 1); usage: bash tools/pmc_loop.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
rm -rf /tmp/pmc_loop
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_loop/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 50 --spinup-ms 0 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("/tmp/pmc_loop/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "bsvi_spec_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    # the timed launch = the dispatch with the largest counters (2000 iterations; the warm-up launch has 50)
    best = max(per.values(), key=lambda d: sum(d.values()))
    for c, v in best.items():
        acc[c] = v
with open("$OUT/pmc_sq_loop.csv", "w") as o:
    o.write("kernel,counter,value_of_the_2000_iteration_launch,per_wave_iteration\n")
    waves = acc.get("SQ_WAVES", 5.0)
    for c, v in sorted(acc.items()):
        o.write('"bsvi_spec_kernel",%s,%.1f,%.2f\n' % (c, v, v / waves / 2000.0))
print(open("$OUT/pmc_sq_loop.csv").read())
PY
