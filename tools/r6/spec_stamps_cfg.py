"""round 6: tools/spec_stamps.py for any scalar-path bench workload (cfg2: Beta-Binomial at 4096 samples, cfg3: T = 200 at 1024):
in-kernel stamps of ONE iteration of the training loop on workgroup 0 / thread 0 (diagnostic build via BSVI_SPEC_DEFINES).
python3 tools/r6/spec_stamps_cfg.py cfg2|cfg3"""
import os
import sys
import time

os.environ["BSVI_SPEC_DEFINES"] = "#define SPEC_DEBUG_STAMPS 1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch                                        # noqa: E402
from brancher_amd import engine, workloads as W     # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
builder, kw, n_samples, opt, okw = {"cfg2": ("build_beta_binomial", dict(n_obs=30), 4096, "SGD", dict(lr=0.1)),
                                    "cfg3": ("build_readme_ar", dict(T=200), 1024, "SGD", dict(lr=1e-4)),
                                    "cfg1": ("build_readme_ar", dict(T=20), 300, "SGD", dict(lr=1e-3))}[which]
api = W.native_api()
c = engine.compile_model(getattr(W, builder)(api, **kw), None, "pathwise")
n_it = 2000
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses, _ = c.train(n_it, n_samples, opt, seed=0, **okw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    s = losses[:12].cpu().numpy()
    names = ["draw", "barrier(d)", "body", "sums", "barrier(a)", "epilogue"]
    d = [s[1]] + [s[i + 1] - s[i] for i in range(1, 6)]
    print("%s mode %s wall %.2f us/it | cycles: total %d = " % (which, c.last_mode, wall * 1e6 / n_it, s[6]) + ", ".join("%s %d" % (n, v) for n, v in zip(names, d))
          + " | epilogue: args+loss %d, own+gsum %d, optimizer %d, publish %d" % (s[7] - s[5], s[8] - s[7], s[9] - s[8], s[6] - s[9])
          + " | geometry %s" % (c.native.engine(n_samples, 2),))
