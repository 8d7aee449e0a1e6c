#!/bin/bash
# Round-2 evidence for the dense-link (BASELINE config 4) and amortised (config 5) paths: bench lines, rocprofv3 kernel
# stats, the launch sequence of one config-5 iteration (side stream off so that durations do not overlap), SQ counters and
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes).  Everything lands in gpurun_out/r2/; copy what is cited
# into profiles/r2/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --workload cfg4 --steps 2000 > $OUT/bench_cfg4.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg4.json
python3 bench.py --workload cfg4 --steps 2000 --estimator blackbox --no-cpu-baseline > $OUT/bench_cfg4_blackbox.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg4_blackbox.json
python3 bench.py --workload cfg5 --steps 500 > $OUT/bench_cfg5.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg5.json
python3 bench.py --workload cfg5 --steps 500 --estimator blackbox --no-cpu-baseline > $OUT/bench_cfg5_blackbox.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg5_blackbox.json
python3 bench.py --workload cfg5 --steps 60 --warmup 5 --samples 2048 --no-cpu-baseline > $OUT/bench_cfg5_whole.json 2>/dev/null; cut -c1-200 $OUT/bench_cfg5_whole.json
cd /tmp && export TMPDIR=/tmp
for wl in cfg4 cfg5; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  cp $(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1) $OUT/${wl}_kernel_stats.csv
  head -8 $OUT/${wl}_kernel_stats.csv | cut -c1-150
done
cd $GRAFT_REPO_ROOT
( echo "# one iteration, launch order, side stream OFF (BSVI_AMORT_OVERLAP=0) so that durations do not overlap; rocprofv3 --kernel-trace"
  BSVI_AMORT_OVERLAP=0 bash tools/prof_cfg5.sh cfg5_serial 2>&1 | grep " us " | head -24 ) > $OUT/cfg5_amortized_launch_sequence.txt
cat $OUT/cfg5_amortized_launch_sequence.txt
bash tools/pmc_cfg5.sh cfg5_sq > /dev/null 2>&1; cp $OUT/cfg5_sq/cfg5_pmc_sq.csv $OUT/cfg5_amortized_pmc_sq.csv; grep "gemm_kernel" $OUT/cfg5_amortized_pmc_sq.csv | head -12
bash tools/pmc_hbm.sh cfg4 --workload cfg4 --steps 5 --warmup 2 > /dev/null 2>&1; cp $OUT/cfg4_hbm.csv $OUT/cfg4_pmc_hbm_traffic.csv; cat $OUT/cfg4_pmc_hbm_traffic.csv
bash tools/pmc_hbm.sh cfg5 --workload cfg5 --steps 4 --warmup 1 > /dev/null 2>&1; cp $OUT/cfg5_hbm.csv $OUT/cfg5_pmc_hbm_traffic.csv; cut -c1-140 $OUT/cfg5_pmc_hbm_traffic.csv
