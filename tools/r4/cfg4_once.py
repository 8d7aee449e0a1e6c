"""round 4: a handful of cfg 4 evaluations and nothing else — the cheapest thing to put under `rocprofv3 --pmc`
(bench.py's spin-up and timing loops make a counter pass take minutes).  python3 tools/r4/cfg4_once.py [n_calls]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from brancher_amd import engine, workloads as W

api = W.native_api()
c = engine.compile_model(W.build_logistic_regression(api, dataset_size=60000, batch_size=512, n_features=784, n_classes=10,
                                                     pixels="uint8", q_scale=0.01), None, "pathwise")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    c.evaluate(1024, seed=1, offset=i)
torch.cuda.synchronize()
print("ok", float(c.out[2]))
