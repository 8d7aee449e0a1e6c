#!/bin/bash
# round 5: the disassembly of one kernel of a built object.  usage: bash tools/r5/kernel_isa.sh <object> <kernel name substring> [out]
OBJ=$1; NAME=$2; OUT=${3:-/tmp/kernel.s}
W=$(mktemp -d); cp "$OBJ" $W/o.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $W/o.o > /dev/null
CO=$(ls $W/o.o.* | grep amdgcn | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$CO" | awk -v n="$NAME" '/^[0-9a-f]+ <.*>:$/ {on = index($0, n) > 0} on' > "$OUT"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | awk -v n="$NAME" '/\.name:/ {on = index($0, n) > 0} on && /(vgpr_count|agpr_count|spill_count|group_segment_fixed_size|private_segment)/' | sort -u
wc -l "$OUT"
rm -rf $W
