#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT; cd $ROOT
timeout 1500 python3 -m pytest tests/test_gpu_amortized.py -x -q -m gpu -k "golden or random_arch or wide" 2>&1 | tail -4 > $OUT/check6.txt
for e in "BSVI_AMORT_FUSE_LIK=0" "BSVI_AMORT_FUSE_LIK=1" "BSVI_AMORT_FUSE_LIK=0" "BSVI_AMORT_FUSE_LIK=1"; do
  echo "== cfg5 $e" >> $OUT/check6.txt
  env $e timeout 600 python3 bench.py --workload cfg5 --steps 100 --warmup 10 --no-cpu-baseline --other-configs off --traffic off 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $OUT/check6.txt 2>&1
done
bash tools/r5/cfg5_timeline.sh > /dev/null 2>&1; cp gpurun_out/r5/cfg5_timeline.txt $OUT/cfg5_timeline_fused.txt
cat $OUT/check6.txt; grep "x6gemm\|lik" $OUT/cfg5_timeline_fused.txt
