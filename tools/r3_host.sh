mkdir -p gpurun_out/r3
python3 tools/host_call_probe.py > gpurun_out/r3/host_call_probe.txt 2>&1
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --other-configs off --traffic off --no-cpu-baseline; done > gpurun_out/r3/bench_short_fast.txt 2>&1
BSVI_FAST_TRAIN=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --other-configs off --traffic off --no-cpu-baseline > gpurun_out/r3/bench_short_slow.txt 2>&1
python3 -m pytest tests/test_gpu_parity.py -x -q -k "graph_replayed or sharded_step or perform_inference or trajectory" 2>&1 | tail -3
cat gpurun_out/r3/host_call_probe.txt
python3 - <<'PY'
import json
for f in ("bench_short_fast", "bench_short_slow"):
    for l in open("gpurun_out/r3/%s.txt" % f):
        if l.startswith("{"):
            d = json.loads(l); print(f, round(d["value"]), round(d["ms_per_step"]*1e3, 3), round(d["device_ms_per_step"]*1e3, 3))
PY
