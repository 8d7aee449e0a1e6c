cd $GRAFT_REPO_ROOT
for d in "" "#define SPEC_PAD_COLUMN 1"; do
  echo "defines: [$d]"
  BSVI_JIT_CACHE=0 BSVI_SPEC_DEFINES="$d" python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from brancher_amd import engine, workloads as W
for n in (300, 128):
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    l, _ = c.train(100, n, "SGD", seed=5, lr=1e-3); torch.cuda.synchronize()
    c.train(2000, n, "SGD", seed=0, lr=1e-3); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); c.train(20000, n, "SGD", seed=0, lr=1e-3); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
    print("n %d: %.3f us per iteration, loss[99] %.6f" % (n, best, float(l[99])))
PY
done
