timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "batched_mvn" 2>&1 | tail -25
python3 - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from brancher_amd import engine, workloads as W
for n_points, n in ((32, 512), (100, 512), (128, 512)):
    c = engine.compile_model(W.build_gp_hyperparameters(W.native_api(), n=n_points, jitter=5e-2), None, "pathwise")
    c.train(5, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.train(50, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    print("gp D=%d N=%d: %.1f us per iteration" % (n_points, n, (time.perf_counter() - t0) / 50 * 1e6))
PY
