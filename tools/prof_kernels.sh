#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py invocation; summaries land in gpurun_out/prof_<tag>/
# usage: tools/prof_kernels.sh <tag> <bench.py args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench.json
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv 2>/dev/null
head -12 $out/kernel_stats.csv | cut -c1-220
