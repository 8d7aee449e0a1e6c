#!/bin/bash
# round 5: cfg 5 per iteration for a list of "VAR=value" settings.  usage: bash tools/r5/cfg5_variants.sh "<setting> ..."
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd $ROOT
for setting in ${1:-BSVI_X6_TN=1}; do
  ( export $setting; python bench.py --workload cfg5 --steps 100 --warmup 10 --other-configs off --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$setting: %.1f us/step' % (l['ms_per_step']*1e3))" ) | tee -a $OUT/cfg5_variants.txt
done
