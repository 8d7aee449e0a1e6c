"""Per-iteration time of the specialised kernel against the chain length T of the README AR model (one workgroup of 256
samples, in-kernel loop): the body is straight-line code of ~130 instructions per time step, so beyond the instruction
cache (64 KB) every instruction is fetched from L2 once per iteration."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brancher_amd import engine, workloads as W   # noqa: E402

api = W.native_api()
for T in [int(x) for x in (sys.argv[1:] or ["10", "20", "40", "60", "80", "100", "140", "200"])]:
    c = engine.compile_model(W.build_readme_ar(api, T=T), None, "pathwise")
    n, iters = 256, 2000
    c.train(50, n, "SGD", lr=1e-4, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c.train(iters, n, "SGD", lr=1e-4, seed=1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("T=%4d  %-10s %8.2f us/it   %6.1f ns per time step" % (T, c.last_mode, us, us * 1e3 / T))
