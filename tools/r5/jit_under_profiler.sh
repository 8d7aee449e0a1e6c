#!/bin/bash
# round 5: the same translation units compiled plain / with the GPU initialised / under rocprofv3 --kernel-trace / under --pmc.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5/jit_hashes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
JIT_HASHES_KEEP=$OUT/co_plain python3 $ROOT/tools/r5/jit_hashes.py plain 2>/dev/null | tail -1 > $OUT/hashes.txt
python3 $ROOT/tools/r5/jit_hashes.py gpu_initialised init_gpu 2>/dev/null | tail -1 >> $OUT/hashes.txt
export JIT_HASHES_KEEP=$OUT/co_profiled
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p1 -- python3 $ROOT/tools/r5/jit_hashes.py rocprofv3_kernel_trace init_gpu 2>/dev/null | grep '^{' | tail -1 >> $OUT/hashes.txt
unset JIT_HASHES_KEEP
timeout 300 rocprofv3 --pmc SQ_WAVES -d $OUT/p2 -- python3 $ROOT/tools/r5/jit_hashes.py rocprofv3_pmc init_gpu 2>/dev/null | grep '^{' | tail -1 >> $OUT/hashes.txt
rm -rf $OUT/p1 $OUT/p2
cat $OUT/hashes.txt
