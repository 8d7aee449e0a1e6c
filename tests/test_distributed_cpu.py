"""N>1 path on CPU (gloo, world_size 2): the sample-shard partition and the one all-reduce of
un-normalised sums.  The per-rank sums the HIP kernel would produce are computed here by the
oracle on each rank's shard; the host logic under test is engine.shard / engine.allreduce_sums
and the out-block layout (include/bsvi.h BSVI_OUT_HEADER)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, Golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brancher_amd import engine, lowering
    from brancher_amd.native import OUT_HEADER
    from oracle.svi_oracle import Oracle
    g = Golden(case)
    model = g.build()
    prog = lowering.lower(model)
    oracle = Oracle(model)
    N = g.N
    base, n_local = engine.shard(N, rank, world)
    noise = {k: v[base:base + n_local] for k, v in g.noise.items()}
    # what bsvi_elbo_fwd_bwd leaves in out_dev for this shard: sums, not means
    oracle.zero_grad()
    value = oracle.elbo(n_local, "pathwise", noise)            # mean over the shard
    (value * n_local).backward()                               # -> d(sum_s f_s)/d theta
    out = torch.zeros(OUT_HEADER + prog.n_params)
    out[0] = float(value.detach()) * n_local
    for par, off, size, _ in prog.parameters:
        grad = oracle.named_parameters()[par.name].grad
        if grad is not None:
            out[OUT_HEADER + off:OUT_HEADER + off + size] = grad.reshape(-1)
    engine.allreduce_sums(out)
    loss = -out[0] / N                                         # bsvi_finalize
    grads = -out[OUT_HEADER:] / N
    if rank == 0:
        named = {par.name: grads[off:off + size].numpy().copy() for par, off, size, _ in prog.parameters}
        out_q.put((float(loss), named, (base, n_local)))
    else:
        out_q.put((None, None, (base, n_local)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["readme_ar_T20_N300", "lognormal_normal_N100"])
def test_two_rank_sample_shards_reproduce_the_reference(case):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = Golden(case)
    shards = sorted(r[2] for r in results)
    assert shards[0][0] == 0 and shards[0][0] + shards[0][1] == shards[1][0] and sum(s[1] for s in shards) == g.N
    loss, grads, _ = next(r for r in results if r[0] is not None)
    ref = float(g.data["loss_pathwise"])
    assert abs(loss - ref) <= 1e-5 * abs(ref)
    ref_grads = g.group("grad_pathwise/")
    scale = max(np.abs(v).max() for v in ref_grads.values())
    for name, gr in ref_grads.items():
        assert np.abs(grads[name].reshape(gr.shape) - gr).max() <= 1e-5 * scale, name


def test_shard_partition_is_exact():
    from brancher_amd import engine
    for n in (1, 7, 300, 8192, 1000003):
        for world in (1, 2, 3, 8):
            if n < world:
                continue
            spans = [engine.shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (b0, n0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + n0 == b1
            assert spans[-1][0] + spans[-1][1] == n
            sizes = [s[1] for s in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- amortised path (BASELINE config 5): the same sample-axis shards, rows = samples x minibatch ------------------
def _vae_worker(rank, world, port, case, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brancher_amd import amortized, engine
    from brancher_amd.native import OUT_HEADER
    from oracle.vae_oracle import VaeOracle
    g = Golden(case)
    model = g.build()
    prog = amortized.lower_amortized(model, model.posterior_model, "pathwise")
    oracle = VaeOracle(model)
    N, B = g.N, prog.batch_size
    base, n_local = engine.shard(N, rank, world)
    rows, eps = g.data["minibatch/x"][base:base + n_local], g.data["noise/z"][base:base + n_local]
    # what bsvi_amort_fwd_bwd leaves in out_dev for this shard: sums over its rows, with the GLOBAL log(N) constant
    t = oracle.terms(rows, eps)
    f = t["lp"] + t["H"] - float(np.log(n_local)) + float(np.log(N))
    for p in oracle.named_parameters().values():
        p.grad = None
    f.sum().backward()
    out = torch.zeros(OUT_HEADER + prog.n_params)
    out[0] = float(f.sum().detach())
    enc_link, dec_link = prog.links
    named = oracle.named_parameters()
    by_par = {}
    for tag, link in (("enc", enc_link), ("dec", dec_link)):
        for pname, par in link.named.items():
            by_par[id(par)] = named["%s/%s" % (tag, pname)]
    for par, off, size, _ in prog.parameters:
        out[OUT_HEADER + off:OUT_HEADER + off + size] = by_par[id(par)].grad.reshape(-1)
    engine.allreduce_sums(out)
    loss = -out[0] / (N * B)                                   # bsvi_finalize_step with n = N * B
    grads = -out[OUT_HEADER:] / (N * B)
    if rank == 0:
        res = {}
        for tag, link in (("enc", enc_link), ("dec", dec_link)):
            for pname, par in link.named.items():
                off = next(o for p_, o, _, _ in prog.parameters if p_ is par)
                res["%s/%s" % (tag, pname)] = grads[off:off + par.size].numpy().copy()
        out_q.put((float(loss), res))
    else:
        out_q.put((None, None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_shards_of_the_amortised_path():
    case = "vae_P150_H136_H40_DS40_B16_N9"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vae_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = Golden(case)
    loss, grads = next(r for r in results if r[0] is not None)
    ref = float(g.data["loss_pathwise"])
    assert abs(loss - ref) <= 1e-5 * abs(ref)
    ref_grads = g.group("grad_pathwise/")
    scale = max(np.abs(v).max() for v in ref_grads.values())
    for name, gr in ref_grads.items():
        assert np.abs(grads[name].reshape(gr.shape) - gr).max() <= 1e-5 * scale, name


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver may start it): the process becomes the
    launcher, starts one rank per GPU through torch.distributed.run on 127.0.0.1 and relays their output.  Here the
    ranks only rendezvous (--launch-check, gloo): the GPU step behind the same entry is covered by the -m gpu suite."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone reports
    assert lines[0]["world"] == 2 and lines[0]["ranks"] == 2.0 and lines[0]["rank_sum"] == 1.0
    # a mismatch between --gpus and the launcher's world size is refused before anything else happens
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)
