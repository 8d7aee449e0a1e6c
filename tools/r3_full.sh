mkdir -p gpurun_out/r3
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3/pytest_gpu.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3/bench_driver_like.json 2> gpurun_out/r3/bench_driver_like.err
tail -c 1200 gpurun_out/r3/pytest_gpu.log
tail -c 600 gpurun_out/r3/bench_driver_like.err
python - <<'PY'
import json
line = [l for l in open("gpurun_out/r3/bench_driver_like.json") if l.startswith("{")]
if line:
    d = json.loads(line[-1])
    print("headline", d["value"], d["ms_per_step"], d["device_ms_per_step"], d["roofline"].get("traffic"), d["roofline"].get("traffic_note"))
    for k, v in d.get("other_configs", {}).items():
        print(k, v.get("error") or (v["ms_per_step"], v["device_ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"]))
PY
