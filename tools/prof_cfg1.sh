# kernel-trace stats of the default bench (persistent) and of the stepwise mode; usage: bash tools/prof_cfg1.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_default_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stepwise -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --mode stepwise --steps 5000 > $OUT/bench_stepwise.json 2>/dev/null
find $OUT/prof_default -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-150
find $OUT/prof_stepwise -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-150
cd $GRAFT_REPO_ROOT && python3 bench.py > $OUT/bench_default.json 2>/dev/null; cut -c1-200 $OUT/bench_default.json
