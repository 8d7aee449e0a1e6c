"""Diagnostic: where one fused ELBO launch spends its cycles (in-kernel s_memtime stamps of
workgroup 0) and what clock the chip holds.  Usage: python tools/phase_stamps.py [workload] [N]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brancher_amd import engine, native, workloads as W
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
builder, kwargs, n, opt, okw, desc = WORKLOADS[name]
if len(sys.argv) > 2:
    n = int(sys.argv[2])
c = engine.compile_model(getattr(W, builder)(W.native_api(), **kwargs), None, "pathwise")
lib = native.load()
stamps = torch.zeros(40, dtype=torch.int64, device="cuda")
for _ in range(200):
    c.evaluate(n, seed=0)
torch.cuda.synchronize()
lib.bsvi_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
acc = None
reps = 50
for _ in range(reps):
    c.evaluate(n, seed=0)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    s = raw[:10].reshape(5, 2).astype("float64")
    d = s[1:] - s[:-1]
    acc = d if acc is None else acc + d
lib.bsvi_debug_set_stamps(None)
acc /= reps
tot = acc.sum(0)
clock_ghz = tot[0] / (tot[1] * 10.0) if tot[1] else float("nan")   # memrealtime ticks at 100 MHz
print(desc, "N=%d" % n, c.native.geometry(n), c.program.summary())
for label, (cyc, rt) in zip(("prologue", "forward", "backward", "reduction"), acc):
    print("%-10s %10.0f cycles  %8.2f us" % (label, cyc, rt / 100.0))
print("total      %10.0f cycles  %8.2f us   in-kernel clock %.2f GHz" % (tot[0], tot[1] / 100.0, clock_ghz))

per = raw[10:37].astype("float64")
print("per-instruction forward cycles (first 27):", [int(x) for x in (per[1:] - per[:-1])], "first:", int(per[0] - raw[2]))
