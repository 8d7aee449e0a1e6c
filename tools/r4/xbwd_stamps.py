"""round 4: cycle stamps of two workgroups of dense_xbwd (prologue, then per column block: start, after the product loop, after
the redraw + reduction).  python tools/r4/xbwd_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from brancher_amd import engine, native, workloads as W

api = W.native_api()
c = engine.compile_model(W.build_logistic_regression(api, dataset_size=60000, batch_size=512, n_features=784, n_classes=10,
                                                     pixels="uint8", q_scale=0.01), None, "pathwise")
for _ in range(5):
    c.evaluate(1024, seed=1)
stamps = torch.zeros(16 * 32, dtype=torch.int64, device="cuda")
c.lib.bsvi_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
c.evaluate(1024, seed=1)
torch.cuda.synchronize()
c.lib.bsvi_debug_set_stamps(None)
t = stamps.cpu().numpy().reshape(16, 32)
for w in range(16):
    row = t[w][t[w] > 0]
    if len(row) < 2:
        continue
    d = np.diff(row)
    role = "product" if (w % 8) < 4 else "partner"
    # product waves: staging | then per block (product loop, hand-over); partner waves: staging | then per block (wait, epilogue)
    print("block %s wave %d %s: staging %6d | " % ("0  " if w < 8 else "133", w % 8, role, d[0]) +
          " ".join("[%6d %6d]" % tuple(d[i:i + 2]) for i in range(1, len(d) - 1, 2)), "total", row[-1] - row[0])
