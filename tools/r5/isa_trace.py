"""round 5: a compact trace of the longest loop of a disassembled kernel (tools/r5/kernel_isa.sh): runs of VALU (V), MFMA (M),
LDS reads / writes (R / W), global loads (G), waits and barriers in program order.  usage: python tools/r5/isa_trace.py kernel.s"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
ins = []
for l in lines:
    m = re.match(r"\s+(\w+)\s*(.*?)\s*//\s+([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
best, best_score = None, -1
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(args.split()[0])
        if off > 32767:
            span = (65536 - off) * 4
            # the loop with the most matrix instructions (else the longest)
            score = sum(1 for b, o, _ in ins if a - span <= b <= a and o.startswith("v_mfma")) * 100000 + span
            if score > best_score:
                best, best_score = (a, span), score
lo, hi = best[0] - best[1], best[0]
out, run, kind = [], 0, None


def flush():
    global run, kind
    if run:
        out.append("%s%d" % (kind, run))
    run, kind = 0, None


def cls(op):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "R"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "W"
    if op.startswith("global_load") or op.startswith("buffer_load"):
        return "G"
    if op.startswith("global_store") or op.startswith("buffer_store"):
        return "S"
    return None


for a, op, args in ins:
    if not (lo <= a <= hi):
        continue
    k = cls(op)
    if k:
        if k != kind:
            flush()
            kind = k
        run += 1
    elif op.startswith("s_waitcnt") or op == "s_barrier":
        flush()
        out.append(op.replace("s_waitcnt", "wait") + ("(" + args + ")" if args else ""))
flush()
print("loop of %d bytes: %d instructions" % (best[1], sum(1 for a, _, _ in ins if lo <= a <= hi)))
print(" ".join(out))
