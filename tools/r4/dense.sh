#!/bin/bash
# round 4: the fused exact-data launches of the dense path — tests, cfg 4 bench line, kernel trace.  usage: bash tools/r4/dense.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4/${1:-dense}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_dense_fused.py -x -q 2>&1 | tail -15 > $OUT/tests_fused.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dense or logreg" 2>&1 | tail -8 > $OUT/tests_dense.txt
timeout 300 python bench.py --workload cfg4 --steps 200 --warmup 20 --no-cpu-baseline --traffic off > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
BSVI_DENSE_FUSED=0 timeout 300 python bench.py --workload cfg4 --steps 200 --warmup 20 --no-cpu-baseline --traffic off > $OUT/bench_cfg4_six.json 2>> $OUT/bench_cfg4.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --workload cfg4 --steps 100 --warmup 10 --no-cpu-baseline --traffic off > /dev/null 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg4_kernel_stats.csv
rm -rf $OUT/prof
cat $OUT/tests_fused.txt $OUT/tests_dense.txt; cat $OUT/bench_cfg4.json $OUT/bench_cfg4_six.json | cut -c1-600; head -12 $OUT/cfg4_kernel_stats.csv | cut -c1-150
