"""round 5: compact traces of every loop of a disassembled kernel whose span lies in [lo, hi] bytes (tools/r5/kernel_isa.sh)."""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
lo_span, hi_span = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
ins = []
for l in lines:
    m = re.match(r"\s+(\w+)\s*(.*?)\s*//\s+([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
loops = []
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(args.split()[0])
        if off > 32767:
            loops.append((a, (65536 - off) * 4))


def cls(op):
    for pre, k in (("v_mfma", "M"), ("v_", "V"), ("ds_read", "R"), ("ds_write", "W"), ("global_load", "G"), ("buffer_load", "G"), ("global_store", "S")):
        if op.startswith(pre):
            return k


for a, span in loops:
    if not (lo_span <= span <= hi_span):
        continue
    out, run, kind = [], 0, None
    for b, op, args in ins:
        if not (a - span <= b <= a):
            continue
        k = cls(op)
        if k:
            if k != kind:
                if run:
                    out.append("%s%d" % (kind, run))
                kind, run = k, 0
            run += 1
        elif op.startswith("s_waitcnt") or op == "s_barrier":
            if run:
                out.append("%s%d" % (kind, run))
            run, kind = 0, None
            out.append(op.replace("s_waitcnt", "wait") + ("(" + args + ")" if args else ""))
    if run:
        out.append("%s%d" % (kind, run))
    print(hex(a), span, " ".join(out)[:2500])
    print()
