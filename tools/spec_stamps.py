"""In-kernel time stamps of ONE iteration of the specialised kernel's training loop (diagnostic build: BSVI_SPEC_DEFINES
adds SPEC_DEBUG_STAMPS; the stamps overwrite the first entries of the loss curve).  Cycles of s_memtime relative to the top
of the iteration, on workgroup 0 / thread 0:  1 noise drawn, 2 past the barrier before the body, 3 body done, 4 wave sums
stored, 5 past the barrier after the body, 6 epilogue done."""
import os
import sys
import time

os.environ["BSVI_SPEC_DEFINES"] = "#define SPEC_DEBUG_STAMPS 1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                        # noqa: E402
from brancher_amd import engine, workloads as W     # noqa: E402

api = W.native_api()
c = engine.compile_model(W.build_readme_ar(api, T=20), None, "pathwise")
n_it = 20000
n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 300      # 64: one wave, thread 0 is an owner and waits for nobody
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses, _ = c.train(n_it, n_samples, "SGD", lr=1e-3, seed=0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    s = losses[:12].cpu().numpy()
    names = ["draw", "barrier(d)", "body", "sums", "barrier(a)", "epilogue"]
    d = [s[1]] + [s[i + 1] - s[i] for i in range(1, 6)]
    print("wall %.2f us/it | cycles: total %d = " % (wall * 1e6 / n_it, s[6]) + ", ".join("%s %d" % (n, v) for n, v in zip(names, d))
          + " | epilogue: args+loss %d, own+gsum %d, optimizer %d, publish %d" % (s[7] - s[5], s[8] - s[7], s[9] - s[8], s[6] - s[9])
          + " | implied clock %.2f GHz" % (s[6] / (wall * 1e6 / n_it) / 1e3))
