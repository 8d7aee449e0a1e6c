#!/bin/bash
# round 5: the program-specialised kernels compiled by the clang bundled with torch (roc-7.0.2, what a plain process runs) against the
# system's (roc-7.2.0, what a process runs under rocprofv3 — or with LD_PRELOAD as here): cfg 1 long loop + driver-like, cfg 2, cfg 3.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/compiler_ab
mkdir -p $OUT
cd $ROOT
export BSVI_JIT_CACHE=0
for who in torch system; do
  if [ $who = system ]; then export LD_PRELOAD="/opt/rocm/lib/libamd_comgr.so.3${LD_PRELOAD:+:$LD_PRELOAD}"; fi
  python -c "from brancher_amd import native; print('$who', native.jit_compiler_identity())" >> $OUT/ab.txt
  for w in "cfg1 20000 50" "cfg1 20 5" "cfg2 200 20" "cfg3 200 20"; do
    set -- $w
    timeout 300 python bench.py --workload $1 --steps $2 --warmup $3 --no-cpu-baseline --traffic off --other-configs off 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); print('$who $1 steps=$2: %.3f us/step wall, %.3f us device, value %.0f' % (l['ms_per_step']*1e3, l['device_ms_per_step']*1e3, l['value']))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
