"""round 5: where the host time of one 20-iteration training call of cfg 1 goes — the statements of engine._prepare_fast_train's closure
timed one by one (median of 300 one after the other), in a plain process and in one set up the way bench.py sets itself up
(torch.cuda.set_device, config.set_device("cuda:0"), torch.distributed imported).  usage: python tools/r5/call_anatomy.py [bench]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    import torch.distributed as dist  # noqa: F401
    torch.cuda.set_device(0)
    from brancher_amd import config
    config.set_device("cuda:0")
from brancher_amd import engine, native, workloads as W  # noqa: E402
from brancher_amd.engine import ElboArgs, shared_seed  # noqa: E402

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
c.train(5, 300, "SGD", seed=0, lr=1e-3)
c.train(20, 300, "SGD", seed=0, lr=1e-3)
torch.cuda.synchronize()
print("device object:", repr(c.device))
cfg = native.make_opt_cfg("SGD", lr=1e-3)
args = ElboArgs.from_buffer_copy(c._elbo_args(300, 300, 0, None, 0, 0))
args.offset_dev = None
fn = c.lib.bsvi_train_persistent2
handle, p_args, p_cfg = c.native.handle, C.byref(args), C.byref(cfg)
params, mask_all, mask_first = (C.c_void_p(t.data_ptr()) for t in (c.params, c.mask_all, c.mask_first))
dev = c.device
marks = {}
P = time.perf_counter
for rep in range(300):
    torch.cuda.synchronize()
    t = [P()]
    K = 20
    Ka = (K + 3) // 4 * 4
    _ = c.params.data_ptr(); t.append(P())
    buf = torch.empty(2 * Ka, device=dev); t.append(P())
    args.seed = 0; args.offset = c.iteration; t.append(P())
    args.stream = torch.cuda.current_stream(dev).cuda_stream; t.append(P())
    c.iteration += K
    rc = fn(handle, p_args, p_cfg, params, None, mask_all, mask_first, 0, K, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * Ka)); t.append(P())
    out = buf[:K], buf[Ka:Ka + K]; t.append(P())
    for i, name in enumerate(("params.data_ptr", "torch.empty", "args stores", "current_stream", "library call", "views")):
        marks.setdefault(name, []).append(t[i + 1] - t[i])
    marks.setdefault("whole", []).append(t[-1] - t[0])
    t0 = P(); c.train(20, 300, "SGD", seed=0, lr=1e-3); marks.setdefault("engine.train(20)", []).append(P() - t0)
for name, v in marks.items():
    v.sort()
    print("%-18s %.2f us" % (name, v[len(v) // 2] * 1e6))
