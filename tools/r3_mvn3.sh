timeout 900 python3 -m pytest tests/test_gpu_mvn.py tests/test_gpu_parity.py -x -q -k "mvn or gp_" 2>&1 | tail -4
python3 - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
from brancher_amd import engine, workloads as W
for n_points, n in ((32, 512), (64, 512), (100, 512), (128, 512)):
    c = engine.compile_model(W.build_gp_hyperparameters(W.native_api(), n=n_points, jitter=5e-2), None, "pathwise")
    c.train(5, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.train(50, n, "Adam", lr=1e-2, seed=1)
    torch.cuda.synchronize()
    print("gp_hyperparameters D=%d number_samples=%d: %.1f us per iteration (base program + bsvi_mvn_kernel + full program + optimizer)" % (n_points, n, (time.perf_counter() - t0) / 50 * 1e6))
PY
