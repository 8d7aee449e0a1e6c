import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import test_gpu_random_models as R
fails = []
for name, fn, rng in (("normal", R.test_random_model_matches_oracle, range(24, 160)),
                      ("generic", R.test_random_generic_model_matches_oracle, range(12, 80)),
                      ("vector", R.test_random_vector_model_matches_oracle, range(16, 100)),
                      ("views", R.test_random_view_model_matches_oracle, range(16, 140))):
    for seed in rng:
        try:
            fn(seed)
        except BaseException as e:
            if type(e).__name__ == "Skipped":
                continue
            fails.append((name, seed, type(e).__name__, str(e)[:300]))
print("failures:", len(fails))
for f in fails[:20]:
    print(f)
