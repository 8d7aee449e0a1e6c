#!/bin/bash
# cfg 5 with the six-piece products on / off
for x in 1 0; do
  BSVI_AMORT_X6=$x python3 bench.py --workload cfg5 --steps 300 --warmup 100 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('X6=$x', 'ms_per_step', d['ms_per_step'], 'value', d['value'])"
done
