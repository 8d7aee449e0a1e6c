#!/bin/bash
# round 6: xgemm_nt_glds_kernel with three LDS stages (BSVI_XGEMM_STAGES=3), with and without the 256-row tile (BSVI_XGEMM_TALL=1):
# the Bayesian neural network at config 4's scale and cfg 5, then the kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
O=$OUT/xgemm_stages_ab.txt; : > $O
line () { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/iteration' % (d['ms_per_step']*1e3))"; }
for e in "BSVI_XGEMM_STAGES=2" "BSVI_XGEMM_STAGES=3" "BSVI_XGEMM_STAGES=3 BSVI_XGEMM_TALL=1" "BSVI_XGEMM_STAGES=2 BSVI_XGEMM_TALL=1"; do
  echo "== $e" >> $O
  env $e timeout 900 python3 -m pytest tests/test_gpu_bnn.py tests/test_gpu_amortized.py -x -q -m gpu -k "bnn or exact or xgemm or golden" 2>&1 | tail -1 >> $O
  for w in bnn_cfg4scale cfg5; do
    env $e timeout 600 python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | line | sed "s/^/$w /" >> $O
  done
done
cd /tmp && export TMPDIR=/tmp
for e in "BSVI_XGEMM_STAGES=2" "BSVI_XGEMM_STAGES=3"; do
  rm -rf /tmp/prof_x
  env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o run -- python3 $ROOT/bench.py --workload bnn_cfg4scale --steps 60 --warmup 5 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
  echo "== kernel trace $e" >> $O
  grep "xgemm" $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) | cut -c1-150 >> $O
done
cat $O
