"""Host cost of the pieces of one CompiledELBO.train() call (microseconds, medians).  usage (GPU box): python3 tools/host_call_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from brancher_amd import engine, native, workloads as W

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
c.train(200, 300, "SGD", lr=1e-3, seed=0)
torch.cuda.synchronize()


def med(fn, reps=300):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6


cfg = native.make_opt_cfg("SGD", lr=1e-3)
print("make_opt_cfg        %.1f" % med(lambda: native.make_opt_cfg("SGD", lr=1e-3)))
print("dist_info + shard   %.1f" % med(lambda: engine.shard(300, *engine.dist_info())))
print("ensure_shares       %.1f" % med(lambda: c.native.ensure_shares(300)))
print("broadcast (no-op)   %.1f" % med(lambda: engine.broadcast_from_rank0(c.params)))
print("training_buffers    %.1f" % med(lambda: engine.training_buffers(20, c.program.n_params, c.device, with_state=False)))
print("_elbo_args          %.1f" % med(lambda: c._elbo_args(300, 300, 0, None, 0, 0)))
loss, fin, _ = engine.training_buffers(1, c.program.n_params, c.device, with_state=False)
args = c._elbo_args(300, 300, 0, None, 0, 0)
ptr = lambda t: C.c_void_p(t.data_ptr())


def launch():
    c.lib.bsvi_train_persistent2(c.native.handle, C.byref(args), C.byref(cfg), ptr(c.params), None, ptr(c.mask_all),
                                 ptr(c.mask_first), 0, 1, ptr(loss), ptr(fin))


def launch_sync():
    launch()
    torch.cuda.synchronize()


print("C call (1 iteration, queue not full: launch + sync) %.1f" % med(launch_sync, 200))
t = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launch()
    t.append(time.perf_counter() - t0)
print("C call alone        %.1f" % (np.median(t) * 1e6))
print("whole train(K=1) host %.1f" % med(lambda: (torch.cuda.synchronize(), c.train(1, 300, "SGD", lr=1e-3, seed=0))[1] and None, 200))
t = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.train(20, 300, "SGD", lr=1e-3, seed=0)
    t.append(time.perf_counter() - t0)
print("train(K=20) host call %.1f" % (np.median(t) * 1e6))
