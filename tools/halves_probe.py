"""Probe: does running the amortised iteration as two concurrent halves of the samples (two streams) hide the last-round
tails of the f32 products?  Times K iterations of one 256-sample call against two concurrent 128-sample calls."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from brancher_amd import engine, native, workloads as W  # noqa: E402

api = W.native_api()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = lambda: W.build_vae(api, dataset_size=60000, batch_size=100, n_features=784, hidden1=512, hidden2=256, seed=0)
c = [engine.compile_model(model(), None, "pathwise") for _ in range(2)]
c = [getattr(x, "__wrapped__", x) for x in c]
lib = c[0].lib
K = 400


def run_whole():
    for it in range(K):
        args = c[0]._args(N, N, 0, seed=1, offset=it)
        native.check(lib.bsvi_amort_fwd_bwd(c[0].handle, C.byref(args)))


streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run_halves():
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)
    for it in range(K):
        for h in range(2):
            with torch.cuda.stream(streams[h]):
                args = c[h]._args(N // 2, N, h * (N // 2), seed=1, offset=it)
                native.check(lib.bsvi_amort_fwd_bwd(c[h].handle, C.byref(args)))
    for s in streams:
        cur.wait_stream(s)


for name, fn in (("whole", run_whole), ("halves", run_halves), ("whole", run_whole), ("halves", run_halves)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-7s %8.1f us per iteration" % (name, e0.elapsed_time(e1) * 1000 / K))
