#!/bin/bash
# round 6: the Bayesian neural network after a change — its tests, both bench entries, the kernel trace at config 4's scale
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
O=$OUT/bnn_check.txt; : > $O
timeout 1500 python3 -m pytest tests/test_gpu_bnn.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_specialised.py -x -q -m gpu -k "bnn" 2>&1 | tail -2 >> $O
line () { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/iteration  loss %.4f' % (d['ms_per_step']*1e3, d['final_loss']))"; }
for w in bnn bnn_cfg4scale bnn bnn_cfg4scale; do
  echo "== $w" >> $O
  timeout 600 python3 bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | line >> $O
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o run -- python3 $ROOT/bench.py --workload bnn_cfg4scale --steps 60 --warmup 5 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
head -10 $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) | cut -c1-140 >> $O
cat $O
