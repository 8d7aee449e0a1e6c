"""Round 6 diagnostic (BSVI_X6_DEBUG=9): the timeline of the workgroups of one x6gemm_kernel launch — s_memtime stamps of one lane per
workgroup (start, top of step 0 / 1 / last, loop end, epilogue end), the 100 MHz wall clock at the start, the hardware id.
BSVI_X6_DEBUG=9 python3 tools/r6/x6_stamps.py [N K]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from brancher_amd import native

lib = native.load()
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
M = 25600
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
A, Bnt, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
Cm = torch.zeros(M, N, device=dev)
for _ in range(20):
    native.check(lib.bsvi_debug_gemm(5, ptr(A), ptr(Bnt), ptr(Cm), None, M, N, K, K, K, N, ptr(bias), 0, 1, 0.0, 0, None))
torch.cuda.synchronize()
tiles = ((M + 127) // 128) * ((N + 127) // 128)
st = torch.zeros(8 * tiles, dtype=torch.int64, device=dev)
native.check(lib.bsvi_debug_gemm(8, None, None, ptr(st), None, 8 * tiles, 0, 0, 0, 0, 0, None, 0, 0, 0.0, 0, None))
s = st.cpu().numpy().reshape(tiles, 8).astype(np.float64)
wall0 = s[:, 6].min()
start_us = (s[:, 6] - wall0) / 100.0
cyc = s[:, :6] - s[:, :1]
n_steps = (K + 31) // 32
print("N %d K %d: %d tiles, %d steps" % (N, K, tiles, n_steps))
first = start_us < 1.0
print("workgroups starting within 1 us of the first: %d; later ones start at %.1f .. %.1f us (median %.1f)" % (
    first.sum(), start_us[~first].min() if (~first).any() else 0, start_us[~first].max() if (~first).any() else 0, np.median(start_us[~first]) if (~first).any() else 0))
for name, sel in (("first round", first), ("later", ~first)):
    if not sel.any():
        continue
    c = cyc[sel]
    med = lambda x: np.median(x)
    print("%-12s cycles (median): start -> step 0 %6.0f | step 0 -> 1 %6.0f | per middle step %6.0f | last step %6.0f | epilogue %6.0f | whole %6.0f" % (
        name, med(c[:, 1]), med(c[:, 2] - c[:, 1]), med((c[:, 3] - c[:, 2]) / max(n_steps - 2, 1)), med(c[:, 4] - c[:, 3]), med(c[:, 5] - c[:, 4]), med(c[:, 5])))
    print("%-12s p10 / p90 of whole: %6.0f / %6.0f cycles" % (name, np.percentile(c[:, 5], 10), np.percentile(c[:, 5], 90)))
# clock: cycles per 10 ns tick need two wall stamps; the whole kernel in wall time from the starts of the last workgroups
print("last start %.1f us after the first" % start_us.max())
