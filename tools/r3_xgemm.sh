mkdir -p gpurun_out/r3
timeout 600 python3 -m pytest tests/test_gpu_collective.py -x -q 2>&1 | tail -8
timeout 900 python3 -m pytest tests/test_gpu_amortized.py -x -q 2>&1 | tail -8
python3 bench.py --workload cfg5 --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline > gpurun_out/r3/bench_cfg5_xgemm.json 2>gpurun_out/r3/bench_cfg5_xgemm.err
BSVI_AMORT_XGEMM=0 python3 bench.py --workload cfg5 --steps 50 --warmup 5 --other-configs off --traffic off --no-cpu-baseline > gpurun_out/r3/bench_cfg5_f32.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("bench_cfg5_xgemm", "bench_cfg5_f32"):
    for l in open("gpurun_out/r3/%s.json" % f):
        if l.startswith("{"):
            d = json.loads(l); print(f, round(d["ms_per_step"]*1e3, 1), round(d["device_ms_per_step"]*1e3, 1), d["roofline"]["achieved"])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 30 --warmup 3 --other-configs off --traffic off --no-cpu-baseline --spinup-ms 0 > /dev/null 2>&1
f=$(find /tmp/prof_cfg5 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r3/cfg5_xgemm_kernel_stats.csv; head -12 $f | cut -c1-150
