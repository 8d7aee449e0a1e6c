mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 30 --warmup 3 --other-configs off --traffic off --no-cpu-baseline --spinup-ms 0 > /dev/null 2>&1
f=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r3/cfg4_exact_kernel_stats.csv; head -12 $f | cut -c1-140
