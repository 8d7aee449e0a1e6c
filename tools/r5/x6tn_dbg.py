import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from brancher_amd import native
lib = native.load(); dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
M = N = 128; K = 32
for krow in (0, 1, 7, 8, 15, 16, 17, 31):
    A = torch.zeros(K, M, device=dev); A[krow, :] = 1.0
    B = torch.arange(K, device=dev, dtype=torch.float32).reshape(K, 1).repeat(1, N) + 1.0      # B[k][n] = k + 1
    Cm = torch.zeros(M, N, device=dev)
    native.check(lib.bsvi_debug_gemm(7, ptr(A), ptr(B), ptr(Cm), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, None))
    torch.cuda.synchronize()
    print("A row", krow, "-> C values", sorted(set(Cm.reshape(-1).tolist()))[:6], "expected", krow + 1)
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); Cm = torch.zeros(M, N, device=dev)
native.check(lib.bsvi_debug_gemm(7, ptr(A), ptr(B), ptr(Cm), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, None))
ref = A.double().T @ B.double()
err = (Cm.double() - ref).abs()
print("random: max err", err.max().item(), "at", divmod(int(err.argmax()), N), "rows with err>1e-3:", sorted(set((err > 1e-3).nonzero()[:, 0].tolist()))[:20], "cols:", sorted(set((err > 1e-3).nonzero()[:, 1].tolist()))[:20])
