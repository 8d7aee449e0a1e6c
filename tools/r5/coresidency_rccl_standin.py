"""round 5: a float4 sum over rank buffers with packed-f32 instructions (tools/r5/pk_victim.hip — a stand-in for
RCCL's reduction kernel, which is not ours to recompile) beside every bf16-MFMA product of the library on a second stream: the
constellation of cfg 5's all-reduce overlapped with the backward pass.  Bits of the sums against the sums taken alone.
python3 tools/r5/coresidency_rccl_standin.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from brancher_amd import native

lib = native.load()
victim = C.CDLL(os.path.join(ROOT, "tools", "bin", "libpkvictim.so"))
victim.pk_victim_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
RANKS, N = 8, 668948 // 4 * 4                # cfg 5's message: 2.7 MB per rank
bufs = torch.randn(RANKS, N, device=dev)


def reduce_into(out, stream):
    assert victim.pk_victim_sum(ptr(bufs), ptr(out), N, RANKS, 4, C.c_void_p(stream.cuda_stream)) == 0


s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
solo, again = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
with torch.cuda.stream(s_side):
    reduce_into(solo, s_side)
    reduce_into(again, s_side)
torch.cuda.synchronize()
print("alone, repeat identical:", bool(torch.equal(solo, again)))

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


bad_total = 0
for kind in ("nothing", "f32 input-gradient (mode 1)", "x6 input-gradient (mode 6)", "x6 forward (mode 5)", "exact-data bf16 x3 (mode 3)"):
    outs = [torch.zeros(N, device=dev) for _ in range(24)]
    torch.cuda.synchronize()
    for o in outs:
        if kind != "nothing":
            with torch.cuda.stream(s_main):
                partner(kind, s_main)
                partner(kind, s_main)
        with torch.cuda.stream(s_side):
            reduce_into(o, s_side)
    torch.cuda.synchronize()
    bad = [int((o != solo).sum()) for o in outs]
    worst = max(float((o - solo).abs().max()) for o in outs)
    bad_total += sum(1 for b in bad if b)
    print("%-32s launches with differing values %2d / %d, values differing (max per launch) %6d of %d, largest difference %.3g" % (
        kind, sum(1 for b in bad if b), len(outs), max(bad), N, worst))
sys.exit(1 if bad_total else 0)
