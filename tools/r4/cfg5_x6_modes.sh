#!/bin/bash
# cfg 5 with the six-piece products in the forward pass only (BSVI_X6_MODES=1, the default) / also for the input gradients (3) / off
for m in 1 3 1 3; do
  BSVI_X6_MODES=$m python3 bench.py --workload cfg5 --steps 300 --warmup 100 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('BSVI_X6_MODES=$m', 'ms_per_step', d['ms_per_step'], 'value', d['value'])"
done
