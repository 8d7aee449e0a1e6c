#!/bin/bash
# round 6: dense_xfwd after a change — the dense tests, cfg 4 per iteration (three runs), the kernel trace, the stamps
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
O=$OUT/xf_check.txt; : > $O
timeout 1500 python3 -m pytest tests/test_gpu_dense_fused.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_specialised.py -x -q -m gpu -k "dense or logistic" 2>&1 | tail -2 >> $O
line () { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/iteration  loss %.4f' % (d['ms_per_step']*1e3, d['final_loss']))"; }
for rep in 1 2 3; do
  timeout 600 python3 bench.py --workload cfg4 --steps 200 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | line >> $O
done
timeout 600 python3 bench.py --workload cfg4 --estimator blackbox --steps 200 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | line >> $O
BSVI_XF_DEBUG=6 timeout 300 python3 tools/r6/xfwd_stamps.py 2>/dev/null | head -8 | cut -c1-330 >> $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 100 --warmup 10 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
head -5 $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) | cut -c1-140 >> $O
cat $O
