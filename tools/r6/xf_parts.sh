#!/bin/bash
# round 6: dense_xfwd taken apart (BSVI_XF_DEBUG: 1 constants drawn, 2 no products, 3 no epilogue, 4 = 1 + 2: the DMA alone, 5 no DMA)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
O=$OUT/xf_parts.txt; : > $O
timeout 1500 python3 -m pytest tests/test_gpu_dense_fused.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or logistic" 2>&1 | tail -2 >> $O
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 3 4 5; do
  rm -rf /tmp/prof_x
  BSVI_XF_DEBUG=$e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 100 --warmup 10 --spinup-ms 0 --no-cpu-baseline --other-configs off --traffic off > /dev/null 2>&1
  echo "== BSVI_XF_DEBUG=$e" >> $O
  grep "dense_xfwd" $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) | cut -c1-120 >> $O
done
cat $O
