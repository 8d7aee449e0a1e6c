"""Times the REAL reference (LucaAmbrogioni/Brancher at /root/reference, PyTorch-CPU) on BASELINE config 1 in the build
container and writes tests/golden/reference_cpu_timings.json.  The reference never travels to the GPU box; bench.py reports
these numbers, labelled with the hardware they were taken on, as cpu_baseline["reference"] beside the oracle port it times
live.  Test infrastructure (see the header of svi_oracle.py).
usage: PYTHONPATH=/root/reference:/root/repo python oracle/time_reference.py"""
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "reference_cpu_timings.json")


def one_run(threads, iters):
    code = r'''
import sys, time, warnings
warnings.filterwarnings("ignore")
import numpy as np, torch
torch.set_num_threads(%d)
sys.path.insert(0, "/root/reference"); sys.path.insert(0, %r)
import oracle.gen_golden as G
api = G.reference_api()
import brancher_amd.workloads as W
from brancher import inference, gradient_estimators as ge
from brancher.optimizers import ProbabilisticOptimizer
model = W.build_readme_ar(api, T=20)
model.update_observed_submodel()
q = model.posterior_model
method = inference.ReverseKL(gradient_estimator=ge.PathwiseDerivativeEstimator)
opts = [ProbabilisticOptimizer(q, "SGD", lr=1e-3)]
def step():
    loss = method.compute_loss(model, q, None, 300)
    [o.zero_grad() for o in opts]
    loss.backward()
    opts[0].update()
for _ in range(3): step()
t0 = time.perf_counter()
for _ in range(%d): step()
print((time.perf_counter() - t0) / %d)
''' % (threads, ROOT, iters, iters)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
    return float(out.stdout.strip().split("\n")[-1])


if __name__ == "__main__":
    import torch
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    rows = []
    for threads in (1, 8):
        s = one_run(threads, 60)
        rows.append(dict(threads=threads, seconds_per_iteration=s, iters_per_sec=1.0 / s, iterations_timed=60))
        print(threads, "threads:", 1.0 / s, "it/s")
    json.dump(dict(workload="README AR state-space T=20, number_samples=300, SGD lr=1e-3 (BASELINE config 1): the reference's own "
                            "compute_loss -> backward -> ProbabilisticOptimizer.update loop (inference.py:95-108)",
                   reference="LucaAmbrogioni/Brancher @ /root/reference, PyTorch-CPU", torch=torch.__version__,
                   hardware=dict(cpu=cpu, logical_cpus=os.cpu_count(), machine=platform.machine(), where="build container (no GPU)"),
                   measured_unix_time=int(time.time()), runs=rows), open(OUT, "w"), indent=1)
    print("wrote", OUT)
