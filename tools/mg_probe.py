"""Diagnostic: cost per iteration of the multi-GPU step sequence, run on ONE GPU with a world-size-1 RCCL group:
bsvi_elbo_fwd_bwd -> all_reduce -> bsvi_finalize_step[_counted], launched eagerly from Python and replayed from HIP graphs
(engine.CompiledELBO._train_graph), against the single-GPU in-kernel loop.  Host time = time to enqueue; total = until done."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29513", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from brancher_amd import config, engine, workloads as W   # noqa: E402

config.set_device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
engine.dist_info = lambda: (0, 1)
# a world-size-1 all_reduce is skipped by allreduce_sums: force the collective so that its launch cost is in the numbers
engine.allreduce_sums = lambda out: (dist.all_reduce(out, op=dist.ReduceOp.SUM), out)[1]


def timed(label, **kw):
    c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
    c.train(64, 300, "SGD", lr=1e-3, seed=0, **kw)
    torch.cuda.synchronize()
    t = time.perf_counter()
    losses, _ = c.train(K, 300, "SGD", lr=1e-3, seed=0, **kw)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print("%-34s mode %-16s host %6.1f us/iter  total %6.1f us/iter  final loss %.4f"
          % (label, c.last_mode, t_host * 1e6 / K, t_all * 1e6 / K, float(losses[-1])))


timed("single GPU, in-kernel loop")
timed("sharded step, HIP graph replay", _force_sharded_path=True)
os.environ["BSVI_GRAPH_UNROLL"] = "1"
timed("sharded step, graph of ONE step", _force_sharded_path=True)
os.environ["BSVI_GRAPH"] = "0"
timed("sharded step, eager from Python", _force_sharded_path=True)
dist.destroy_process_group()
