"""round 6: where the host time of a repeated README perform_inference call goes (cProfile, cumulative), after the code object is in memory
python3 tools/r6/perform_inference_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from brancher_amd import inference, workloads as W

api = W.native_api()
for i in range(3):
    model = W.build_readme_ar(api, T=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inference.perform_inference(model, number_iterations=500, number_samples=300, optimizer="SGD", lr=0.001)
    loss = model.diagnostics["loss curve"]
    print("call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
model = W.build_readme_ar(api, T=20)
pr = cProfile.Profile()
pr.enable()
inference.perform_inference(model, number_iterations=500, number_samples=300, optimizer="SGD", lr=0.001)
loss = model.diagnostics["loss curve"]
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:9000])
# the same model object again (what a script that calls perform_inference twice does)
for i in range(3):
    t0 = time.perf_counter()
    inference.perform_inference(model, number_iterations=500, number_samples=300, optimizer="SGD", lr=0.001)
    loss = model.diagnostics["loss curve"]
    print("same model, call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
