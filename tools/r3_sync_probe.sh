#!/bin/bash
# does the way the host waits for the GPU (interrupt or polling) change the driver-timed headline?
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2 3; do
  for mode in default nointr; do
    if [ $mode = nointr ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', round(d['value']), d['ms_per_step']*1e3, d.get('device_ms_per_step',0)*1e3)"
  done
done
