"""round 5: the Bayesian neural network (bsvi_bnn_*) per training iteration — the reference's example (784-20-10, minibatch 30, 50 samples,
tests/test_MNIST_bayesian_neural_network.py:56-60) and the same network at BASELINE config 4's scale (minibatch 512, 1024 samples).
python3 tools/r5/bnn_timing.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from brancher_amd import engine, workloads as W

for tag, kw, n in (("example: 784-20-10, B 30, N 50", dict(dataset_size=60000, batch_size=30, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4, q_loc_scale=1.0), 50),
                   ("config-4 scale: 784-20-10, B 512, N 1024", dict(dataset_size=60000, batch_size=512, n_features=784, n_hidden=20, n_classes=10, q_scale1=4e-4, q_loc_scale=1.0), 1024)):
    c = engine.compile_model(W.build_bayesian_neural_network(W.native_api(), **kw), None, "pathwise")
    c.train(20, n, "Adam", lr=5e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses, finite = c.train(100, n, "Adam", lr=5e-3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    flops = 2 * 2.0 * n * kw["n_hidden"] * kw["n_features"] * kw["batch_size"]
    print("%s: %.1f us per iteration (%s products, %.1f TFLOP/s on the two products' %.2f GFLOP), loss %.2f -> %.2f, all finite %s" % (
        tag, dt * 1e6, c.data_path(), flops / dt / 1e12, flops / 1e9, float(losses[0]), float(losses[-1]), bool(finite.all())))
