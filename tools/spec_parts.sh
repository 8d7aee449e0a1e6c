#!/bin/bash
# what the parts of the specialised kernel cost (cfg 1, in-kernel loop): the whole, without the noise, without the body
for d in "" "#define SPEC_DEBUG_NO_DRAW 1" "#define SPEC_DEBUG_NO_BODY 1" $'#define SPEC_DEBUG_NO_DRAW 1\n#define SPEC_DEBUG_NO_BODY 1'; do
  echo "== defines: $d"
  BSVI_SPEC_DEFINES="$d" python bench.py --steps 20000 --warmup 100 --no-cpu-baseline --spinup-ms 100 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('us/step', j['device_ms_per_step']*1e3, 'loss', j['final_loss'])"
done
