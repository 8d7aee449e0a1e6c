"""Run a few fused ELBO evaluations (for rocprofv3 counter collection)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brancher_amd import engine, workloads as W
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
builder, kwargs, n0, opt, okw, desc = WORKLOADS[name]
n = n or n0
c = engine.compile_model(getattr(W, builder)(W.native_api(), **kwargs), None, "pathwise")
for _ in range(reps):
    c.evaluate(n, seed=0)
torch.cuda.synchronize()
print("done", name, n, c.native.geometry(n))
