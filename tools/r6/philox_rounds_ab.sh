#!/bin/bash
# round 6: Philox4x32 with 10 rounds (as built) against 7 (the Random123 minimum that passes BigCrush) on the draw-bound kernels:
# cfg 4 (dense_xfwd / dense_xbwd), the BNN at config 4's scale, cfg 1 — rebuilds the library on the box  -> gpurun_out/r6/philox_rounds.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/philox_rounds.txt; : > $OUT
run () {
  echo "== Philox rounds $1" >> $OUT
  for w in cfg4 cfg1; do
    python3 bench.py --workload $w --steps $( [ $w = cfg1 ] && echo 20000 || echo 100 ) --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step']*1e3, 'us', d['value'])" >> $OUT 2>&1
  done
  python3 tools/r5/bnn_timing.py 2>/dev/null | tail -3 >> $OUT
}
run 10
sed -i 's/for (int r = 0; r < 10; ++r) {/for (int r = 0; r < 7; ++r) {/' brancher_amd/csrc/philox.h
make -C brancher_amd/csrc > /dev/null 2>&1
export BSVI_CACHE_DIR=/tmp/bsvi_cache_p7
run 7
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "philox" 2>&1 | tail -3 >> $OUT
cat $OUT
