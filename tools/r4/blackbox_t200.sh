#!/bin/bash
# round 4: the T = 200 programs after the generator's two changes (no SLP vectorizer on long programs; BlackBox sinks in two
# passes) — parity of both engines, cold compile time and time per iteration, old against new.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/bb200
mkdir -p $OUT
cd $ROOT
export BSVI_JIT_CACHE=0
timeout 1500 python3 -m pytest tests/test_gpu_specialised.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
run () {   # tag, env..., -- bench args
  tag=$1; shift
  ( export "$@"; s=$(date +%s.%N); timeout 600 python3 bench.py --workload cfg3 --steps 200 --warmup 5 --no-cpu-baseline --other-configs off --traffic off $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; e=$(date +%s.%N)
    python3 - $OUT/$tag.json $tag $s $e <<'PY'
import json, sys
try:
    l = json.load(open(sys.argv[1]))
    print("%-28s %.2f us/iteration (device %.2f), whole command %.1f s" % (sys.argv[2], l["ms_per_step"] * 1e3, l.get("device_ms_per_step", 0) * 1e3, float(sys.argv[4]) - float(sys.argv[3])))
except Exception as ex:
    print(sys.argv[2], "failed", ex)
PY
  )
}
EXTRA="" run pathwise_new X=1
EXTRA="" run pathwise_slp BSVI_JIT_SLP=1
EXTRA="--estimator blackbox" run blackbox_new X=1
EXTRA="--estimator blackbox" run blackbox_noslp_onepass BSVI_SPEC_TWO_PASS=0
EXTRA="--estimator blackbox" run blackbox_old BSVI_SPEC_TWO_PASS=0 BSVI_JIT_SLP=1
# cfg 1 (short program: the vectorizer stays on) — unchanged source, as a control
python3 bench.py --steps 20000 --warmup 200 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg1 long loop %.3f us' % (l['ms_per_step']*1e3))"
