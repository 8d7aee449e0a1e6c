#!/bin/bash
# the driver's command end to end (wall time of the whole script, headline + other_configs + cpu_baseline), then the two-rank launcher path on one GPU
cd /root/repo
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench wall seconds: $(( $(date +%s) - T0 ))"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/final_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","traffic")})
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"])
print({k: round(v["ms_per_step"]*1e3,1) for k, v in d["other_configs"].items()})
PY
BSVI_BENCH_BACKEND=gloo BSVI_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
