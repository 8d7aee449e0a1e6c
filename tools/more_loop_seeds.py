"""tests/test_gpu_random_models.py::test_random_model_training_loop_equals_launch_per_iteration over many more seeds."""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import test_gpu_random_models as R
fails, ran = [], 0
for family in ("normal", "generic", "vector", "views"):
    for seed in range(6, 46):
        for optimizer, kw in (("SGD", dict(lr=2e-3)), ("Adam", dict(lr=1e-2))):
            try:
                R.test_random_model_training_loop_equals_launch_per_iteration(family, seed, optimizer, kw)
                ran += 1
            except BaseException as e:
                if type(e).__name__ == "Skipped":
                    continue
                fails.append((family, seed, optimizer, type(e).__name__, str(e)[:300]))
print("ran", ran, "failures:", len(fails))
for f in fails[:20]:
    print(f)
