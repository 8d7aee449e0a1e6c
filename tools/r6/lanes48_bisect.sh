#!/bin/bash
# round 6: time-boxed bisect of the "packed-f32 victim returns wrong values in lanes 48-63 beside a bf16-MFMA
# partner" effect of profiles/r4/x6_notes.txt section 4.  A scratch build ON THE BOX re-enables packed f32 in outer_kernel ONLY (a function
# attribute; every other kernel of the library stays without), then tools/r4/coresidency_probe.py runs the victim beside the partners
# — the f32-input MFMA product, the exact-data bf16 x3 product (LDS-DMA), round 5's x6gemm_r5_kernel (register staging, 61 KB LDS, no
# sched_barrier) and round 6's x6gemm_kernel (LDS-DMA of both operands, 80 KB LDS, pinned order) — and beside round 6's kernel with parts
# compiled out (BSVI_X6_DEBUG: 2 no C stores, 3 no MFMAs, 4 every fragment read from one LDS address, 5 no LDS-DMA after the first step).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/r6; OUT=gpurun_out/r6/lanes48_bisect.txt; : > $OUT
sed -i 's/^__global__ __launch_bounds__(256) void outer_kernel(/__global__ __launch_bounds__(256) __attribute__((target("packed-fp32-ops"))) void outer_kernel(/' brancher_amd/csrc/amort_kernel.hip
make -C brancher_amd/csrc > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log >> $OUT; }
W=$(mktemp -d); cp brancher_amd/csrc/build/amort_kernel.o $W/o.o; /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $W/o.o > /dev/null
CO=$(ls $W/o.o.* | grep amdgcn | head -1)
echo "packed-f32 instructions in the scratch build: outer_kernel $(/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO | awk '/^[0-9a-f]+ <.*>:$/ {on = index($0, "outer_kernel") > 0} on' | grep -c 'v_pk_') / rest of the object $(/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO | awk '/^[0-9a-f]+ <.*>:$/ {on = index($0, "outer_kernel") == 0} on' | grep -c 'v_pk_[fam][mdu][adl]_f32')" >> $OUT
for e in "BSVI_X6_V=6" "BSVI_X6_V=5" "BSVI_X6_V=6 BSVI_X6_DEBUG=2" "BSVI_X6_V=6 BSVI_X6_DEBUG=3" "BSVI_X6_V=6 BSVI_X6_DEBUG=4" "BSVI_X6_V=6 BSVI_X6_DEBUG=5" "BSVI_X6_V=6 BSVI_X6_VAR=0"; do
  echo "== $e" >> $OUT
  env $e timeout 300 python3 tools/r4/coresidency_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
