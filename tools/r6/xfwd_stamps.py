"""round 6: cycle stamps of two workgroups of dense_xfwd (BSVI_XF_DEBUG=6), per wave relative to the workgroup's first mark.
Product waves 0-3: start | then (before, after) the products of steps 0..11 | loop end | epilogue start | end.
Drawing waves 4-7: start | then per chunk (drawn, behind the barrier) | epilogue start | end.
BSVI_XF_DEBUG=6 python tools/r6/xfwd_stamps.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from brancher_amd import engine, workloads as W

api = W.native_api()
c = engine.compile_model(W.build_logistic_regression(api, dataset_size=60000, batch_size=512, n_features=784, n_classes=10,
                                                     pixels="uint8", q_scale=0.01), None, "pathwise")
for _ in range(20):
    c.evaluate(1024, seed=1)
stamps = torch.zeros(32 * 32, dtype=torch.int64, device="cuda")
c.lib.bsvi_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
c.evaluate(1024, seed=1)
torch.cuda.synchronize()
c.lib.bsvi_debug_set_stamps(None)
t = stamps.cpu().numpy().reshape(32, 32)[16:]
for blk in range(2):
    rows = t[blk * 8:blk * 8 + 8]
    if not (rows > 0).any():
        print("no stamps (BSVI_XF_DEBUG=6?)")
        break
    t0 = rows[rows > 0].min()
    for w in range(8):
        row = rows[w][rows[w] > 0]
        rel = row - t0
        if w < 4:
            steps = rel[1:25].reshape(-1, 2)
            print("wg %3d product wave %d: start %4d | " % (133 * blk, w, rel[0]) + " ".join("%d+%d" % (a - (steps[i - 1][1] if i else rel[0]), b - a) for i, (a, b) in enumerate(steps)) +
                  " | loop end %d epilogue %d end %d" % (rel[25], rel[26], rel[27]))
        else:
            ch = rel[1:-2].reshape(-1, 2)
            print("wg %3d drawing wave %d: start %4d | " % (133 * blk, w, rel[0]) + " ".join("%d+%d" % (a - (ch[i - 1][1] if i else rel[0]), b - a) for i, (a, b) in enumerate(ch)) +
                  " | epilogue %d end %d" % (rel[-2], rel[-1]))
