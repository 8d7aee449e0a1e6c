cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_big -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload cfg1_big --steps 100 --warmup 10 > /dev/null 2>&1
find /tmp/prof_big -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-140
