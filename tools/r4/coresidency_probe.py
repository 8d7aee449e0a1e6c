"""Does outer_kernel (the narrow layers' weight gradient: packed-f32 VALU, 4 KB LDS) return the same bits when another kernel
runs beside it on a second stream?  Partner kernels: the f32-input MFMA product (bsvi_debug_gemm mode 1), the six-piece bf16
product (modes 5 / 6, x6gemm_kernel), the exact-data bf16 product (mode 3), or nothing.  profiles/r4/x6_notes.txt section 4.
python3 tools/r4/coresidency_probe.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from brancher_amd import native

lib = native.load()
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
K, M, N = 25600, 4, 512                       # dW[4][512] = dY[K][4]^T h[K][512]: the encoder heads of cfg 5
A = torch.randn(K, M, device=dev)
B = torch.randn(K, N, device=dev)
colsum = torch.zeros(M, device=dev)


def outer(out, stream):
    out.zero_()
    native.check(lib.bsvi_debug_gemm(2, ptr(A), ptr(B), ptr(out), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, C.c_void_p(stream.cuda_stream)))


s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
solo = torch.zeros(M, N, device=dev)
with torch.cuda.stream(s_side):
    outer(solo, s_side)
torch.cuda.synchronize()
again = torch.zeros(M, N, device=dev)
with torch.cuda.stream(s_side):
    outer(again, s_side)
torch.cuda.synchronize()
print("solo repeat identical:", bool(torch.equal(solo, again)))

Mg, Ng, Kg = 25600, 256, 512
Ag, Bnn, Bnt = torch.randn(Mg, Kg, device=dev), torch.randn(Kg, Ng, device=dev), torch.randn(Ng, Kg, device=dev)
Yg, Cg, bias = torch.randn(Mg, Ng, device=dev), torch.zeros(Mg, Ng, device=dev), torch.randn(Ng, device=dev)
Xexact = torch.randint(0, 2, (Mg, Kg), device=dev).float()


def partner(kind, stream):
    st = C.c_void_p(stream.cuda_stream)
    if kind == "f32 input-gradient (mode 1)":
        native.check(lib.bsvi_debug_gemm(1, ptr(Ag), ptr(Bnn), ptr(Cg), None, Mg, Ng, Kg, Kg, Ng, Ng, ptr(Yg), Ng, 1, 0.0, 0, st))
    elif kind == "x6 input-gradient (mode 6)":
        native.check(lib.bsvi_debug_gemm(6, ptr(Ag), ptr(Bnn), ptr(Cg), None, Mg, Ng, Kg, Kg, Ng, Ng, ptr(Yg), Ng, 1, 0.0, 0, st))
    elif kind == "x6 forward (mode 5)":
        native.check(lib.bsvi_debug_gemm(5, ptr(Ag), ptr(Bnt), ptr(Cg), None, Mg, Ng, Kg, Kg, Kg, Ng, ptr(bias), 0, 1, 0.0, 0, st))
    elif kind == "exact-data bf16 x3 (mode 3)":
        native.check(lib.bsvi_debug_gemm(3, ptr(Xexact), ptr(Bnt), ptr(Cg), None, Mg, Ng, Kg, Kg, Kg, Ng, ptr(bias), 0, 1, 0.0, 0, st))


for kind in ("nothing", "f32 input-gradient (mode 1)", "x6 input-gradient (mode 6)", "x6 forward (mode 5)", "exact-data bf16 x3 (mode 3)"):
    outs = [torch.zeros(M, N, device=dev) for _ in range(24)]
    torch.cuda.synchronize()
    for o in outs:
        if kind != "nothing":
            with torch.cuda.stream(s_main):
                partner(kind, s_main)
                partner(kind, s_main)
        with torch.cuda.stream(s_side):
            outer(o, s_side)
    torch.cuda.synchronize()
    bad = [int((o != solo).sum()) for o in outs]
    worst = max(float((o - solo).abs().max()) for o in outs)
    cols = sorted({int(i) % N for o in outs for i in torch.nonzero((o != solo).reshape(-1)).reshape(-1).cpu().numpy().tolist()})
    print("%-32s launches with differing values %2d / %d, values differing (max per launch) %4d of %d, largest difference %.3g (scale %.3g)%s" % (
        kind, sum(1 for b in bad if b), len(outs), max(bad), M * N, worst, float(solo.abs().max()),
        ("  columns %d..%d" % (cols[0], cols[-1])) if cols else ""))
