"""round 5: cycle stamps of two workgroups of dense_xbwd, as a timeline: per wave the time of every mark relative to the workgroup's
first one (marks: before staging | after the staging barrier | then per column block: start, after the product loop, after the
redraw + reduction).  python tools/r5/xbwd_stamps.py"""
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
for _ in range(20):
    c.evaluate(1024, seed=1)
stamps = torch.zeros(16 * 32, dtype=torch.int64, device="cuda")
c.lib.bsvi_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
c.evaluate(1024, seed=1)
torch.cuda.synchronize()
c.lib.bsvi_debug_set_stamps(None)
t = stamps.cpu().numpy().reshape(16, 32)
for blk in range(2):
    rows = t[blk * 8:blk * 8 + 8]
    t0 = rows[rows > 0].min() if (rows > 0).any() else 0
    for w in range(8):
        row = rows[w][rows[w] > 0]
        if len(row) < 2:
            continue
        rel = row - t0
        d = np.diff(row)
        print("wg %s wave %d: start %5d staged %6d | " % ("0  " if blk == 0 else "133", w, rel[0], rel[1]) +
              " ".join("[loop %6d epi %6d]" % (d[i + 1], d[i + 2]) for i in range(1, len(d) - 2, 3)), "| end", rel[-1])
