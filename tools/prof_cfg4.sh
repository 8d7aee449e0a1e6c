python -m pytest tests -m gpu -q -x -k "dense or logreg" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/$1/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r2/$1/prof -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-110
