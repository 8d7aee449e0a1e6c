import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from brancher_amd import engine, workloads as W
from oracle.svi_oracle import Oracle
kw = dict(dataset_size=96, batch_size=30, n_features=784, n_hidden=20, n_classes=10, q_scale1=0.05, q_loc_scale=1.0, pixels="unit")
n = 20
for est in ("blackbox",):
    c = engine.compile_model(W.build_bayesian_neural_network(W.native_api(), **kw), None, est)
    res = c.evaluate(n, seed=3, offset=5, want_noise=True, want_indices=True, want_fvalues=True)
    idx = res["indices"].cpu().numpy(); named = c.named_noise(res["noise"].cpu().numpy(), n); mb = {c.program.indices_name: idx.tolist()}
    build = lambda: W.build_bayesian_neural_network(W.native_api(), **kw)
    ex = Oracle(build(), dtype=torch.float64).loss_and_grads(n, est, named, mb)
    r32 = Oracle(build()).loss_and_grads(n, est, named, mb)
    f, f64, f32 = res["f"].cpu().numpy().astype(np.float64), ex["f"].reshape(-1), r32["f"].reshape(-1)
    lq, lq64, lq32 = res["lq"].cpu().numpy().astype(np.float64), ex["lq"].reshape(-1), r32["lq"].reshape(-1)
    print("f   ours-64:", np.abs(f - f64).max(), (f - f64).mean(), " ref32-64:", np.abs(f32 - f64).max(), " f range", f64.min(), f64.max())
    print("lq  ours-64:", np.abs(lq - lq64).max(), (lq - lq64).mean(), " ref32-64:", np.abs(lq32 - lq64).max(), " lq range", lq64.min(), lq64.max())
    print("value ours", float(res["loss"].item()), "from our f,lq:", -np.mean(lq * f + f), "exact", ex["loss"], "ref32", r32["loss"])
