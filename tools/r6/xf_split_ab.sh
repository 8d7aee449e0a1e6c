#!/bin/bash
# round 6: dense_xfwd with the last, partly filled pass of the draw on the product waves (BSVI_XF_SPLIT=0: off) — the dense tests, then
# cfg 4 per iteration for a list of settings, then the two kernels under the kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
O=$OUT/xf_split_ab.txt
timeout 1500 python3 -m pytest tests/test_gpu_dense_fused.py tests/test_gpu_c_abi.py -x -q -m gpu 2>&1 | tail -4 > $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or logistic" 2>&1 | tail -4 >> $O
line () { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/iteration  loss %.4f' % (d['ms_per_step']*1e3, d['final_loss']))"; }
for rep in 1 2; do
for e in "BSVI_XF_SPLIT=0" "BSVI_XF_SPLIT=32" "BSVI_XF_SPLIT=16" ${XF_EXTRA}; do
  echo "== cfg4 $e" >> $O
  env $e timeout 600 python3 bench.py --workload cfg4 --steps 200 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | line >> $O
done
done
cd /tmp && export TMPDIR=/tmp
for e in 0 32; do
  rm -rf /tmp/prof_x
  BSVI_XF_SPLIT=$e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 100 --warmup 10 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
  echo "== kernel trace BSVI_XF_SPLIT=$e" >> $O
  head -4 $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) | cut -c1-140 >> $O
done
cat $O
