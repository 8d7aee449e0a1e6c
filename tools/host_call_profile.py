"""cProfile of the host side of short train() calls (what a 20-iteration timed region pays per call)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brancher_amd import engine, workloads as W

c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
for _ in range(200):
    c.train(20, 300, "SGD", lr=1e-3, seed=0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    c.train(20, 300, "SGD", lr=1e-3, seed=0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
