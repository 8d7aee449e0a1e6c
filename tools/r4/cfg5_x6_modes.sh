#!/bin/bash
# cfg 5 with the six-piece products: the default rule (forward products + the input gradients with K >= 512), forward only
# (BSVI_X6_MODES=1), forward + every input gradient (3), off (BSVI_AMORT_X6=0)
run () { python3 bench.py --workload cfg5 --steps 300 --warmup 100 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
for i in 1 2; do
  run "default rule      "
  BSVI_X6_MODES=1 run "BSVI_X6_MODES=1   "
  BSVI_X6_MODES=3 run "BSVI_X6_MODES=3   "
done
BSVI_AMORT_X6=0 run "BSVI_AMORT_X6=0   "
