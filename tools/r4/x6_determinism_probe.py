"""cfg 5 at its per-GPU size, pathwise: the gradient block of two identical calls (bit equality expected) and of two half
shards against the whole (the property the multi-GPU step relies on).  Used with BSVI_X6_MODES / BSVI_X6_NN_ONLY /
BSVI_AMORT_OVERLAP to find which launches may run beside each other (profiles/r4/x6_notes.txt).
python3 tools/r4/x6_determinism_probe.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from brancher_amd import engine, workloads as W
N, B = 256, 100
model = W.build_vae(W.native_api(), dataset_size=4000, batch_size=B, n_features=784, hidden1=512, hidden2=256, seed=7)
cp = engine.compile_model(model, model.posterior_model, "pathwise")
full = cp.evaluate(N, seed=3, offset=0, want_noise=True, want_indices=True)
eps, rows = full["noise"].cpu().numpy().reshape(N, B, 2), full["indices"].cpu().numpy()
runs = []
for _ in range(2):
    r = cp.evaluate(N, noise=eps, minibatch=rows)
    runs.append(r["grads"].clone())
print("repeat full-vs-full max diff", float((runs[0] - runs[1]).abs().max()))
for par, off, size, _ in cp.program.parameters:
    dd = (runs[0] - runs[1]).abs()[off:off+size]
    if float(dd.max()) > 0:
        print("   repeat diff", par.name, float(dd.max()), "n differing", int((dd > 0).sum()), "of", size)
        print("   indices", torch.nonzero(dd > 0).reshape(-1).cpu().numpy().tolist())
        print("   diffs", [float(x) for x in dd[dd > 0].cpu().numpy()][:60])
half = N // 2
gsum = torch.zeros_like(runs[0])
for h in range(2):
    sl = slice(h * half, (h + 1) * half)
    r = cp.evaluate(half, noise=eps[sl], minibatch=rows[sl])
    gsum += r["grads"] * half
d = (gsum / N - runs[0]).abs()
i = int(d.argmax())
print("halves-vs-full max diff", float(d.max()), "at", i, "scale", float(runs[0].abs().max()), "value", float(runs[0][i]))
for par, off, size, _ in cp.program.parameters:
    print(par.name, off, size, "maxdiff", float(d[off:off+size].max()), "scale", float(runs[0][off:off+size].abs().max()))
