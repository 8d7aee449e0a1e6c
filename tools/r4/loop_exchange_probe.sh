#!/bin/bash
# round 4: what the exchange inside the in-kernel loop costs per iteration — one rank exchanging with itself against the plain
# loop, and two rank processes sharing this box's GPU (protocol overhead only: no xGMI hop) with the loop exchange on and off;
# and cfg 1's kernel with the SLP vectorizer off.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/xloop
mkdir -p $OUT
cd $ROOT
python3 - <<'PY' | tee $OUT/one_rank.txt
import os, sys, time
sys.path.insert(0, ".")
import torch
from brancher_amd import engine, workloads as W
def run(force, n, K=4000):
    if force: os.environ["BSVI_LOOP_EXCHANGE"] = "force"
    else: os.environ.pop("BSVI_LOOP_EXCHANGE", None)
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    kw = dict(_force_sharded_path=True) if force else {}
    c.train(50, n, "SGD", seed=1, lr=1e-3, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); c.train(K, n, "SGD", seed=1, lr=1e-3, **kw); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    return best, c.last_mode
for n in (300, 128):
    a, ma = run(False, n); b, mb = run(True, n)
    print("N=%d: %s %.3f us/iteration, %s %.3f us/iteration (one rank exchanging with itself)" % (n, ma, a, mb, b))
PY
for loop in 1 0; do
  BSVI_LOOP_EXCHANGE=$loop BSVI_BENCH_BACKEND=gloo BSVI_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 2000 --warmup 50 --spinup-ms 0 > $OUT/two_ranks_loop$loop.json 2> $OUT/two_ranks_loop$loop.err
  python3 -c "
import json,sys
l=[json.loads(x) for x in open('$OUT/two_ranks_loop$loop.json') if x.startswith('{')][0]
print('two ranks sharing the GPU, BSVI_LOOP_EXCHANGE=$loop: mode %s, %.2f us per iteration (2000-iteration call)' % (l['config']['mode'], l['ms_per_step']*1e3))" | tee -a $OUT/two_ranks.txt
done
for slp in 1 0; do
  BSVI_JIT_CACHE=0 BSVI_JIT_SLP=$slp python3 bench.py --steps 20000 --warmup 200 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg1 long loop, SLP vectorizer=$slp: %.3f us' % (l['ms_per_step']*1e3))" | tee -a $OUT/slp.txt
done
