"""The N>1 path on REAL kernels: two processes, one rank each, sharing the one GPU of the test box; collectives over gloo
(RCCL refuses two ranks on one device — on a multi-GPU node the same code runs with backend "nccl").  Each rank evaluates
its shard of the Monte-Carlo samples with the HIP kernels, the output blocks are all-reduced, every rank applies the same
optimizer step: the trajectory must be the single-process one (same Philox streams: a sample's draws depend on its global
index, not on the rank that evaluates it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, workload, n_samples, iters, optimizer, opt_kw, out_q, collective="torch", loop_exchange="1"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BSVI_COLLECTIVE"] = collective
    os.environ["BSVI_LOOP_EXCHANGE"] = loop_exchange
    # a gloo collective cannot be captured into a HIP graph; the library's one-shot exchange is a kernel and can
    os.environ["BSVI_GRAPH"] = "1" if collective in ("exchange", "auto") else "0"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brancher_amd import config, engine, workloads as W
    config.set_device("cuda:0")
    torch.manual_seed(1234 + 77 * rank)           # ranks disagree on torch's seed on purpose: rank 0's must win
    builder, kwargs = workload
    model = getattr(W, builder)(W.native_api(), **kwargs)
    c = engine.compile_model(model, None, "pathwise")
    if rank == 1:                                 # ... and on the initial parameters: broadcast from rank 0 must fix it
        c.params.add_(0.5)
    losses, finite = c.train(iters, n_samples, optimizer, seed=11, **opt_kw)
    res = c.evaluate(n_samples, seed=5, offset=900)
    torch.cuda.synchronize()
    out_q.put((rank, losses.cpu().numpy(), c.params.detach().cpu().numpy().copy(), float(res["loss"].item()),
               c.last_mode, bool(finite.all()), bool(engine._exchanges.get(0))))
    dist.barrier()
    dist.destroy_process_group()


def _single(workload, n_samples, iters, optimizer, opt_kw):
    sys.path.insert(0, ROOT)
    from brancher_amd import engine, workloads as W
    builder, kwargs = workload
    model = getattr(W, builder)(W.native_api(), **kwargs)
    c = engine.compile_model(model, None, "pathwise")
    losses, finite = c.train(iters, n_samples, optimizer, seed=11, allow_persistent=False, **opt_kw)
    res = c.evaluate(n_samples, seed=5, offset=900)
    return losses.cpu().numpy(), c.params.detach().cpu().numpy().copy(), float(res["loss"].item())


@pytest.mark.parametrize("workload,n_samples,optimizer,opt_kw", [
    (("build_readme_ar", dict(T=20)), 600, "SGD", dict(lr=1e-3)),
    (("build_logistic_regression", dict(dataset_size=256, batch_size=64, n_features=64, n_classes=10, q_scale=0.05)), 96, "Adam", dict(lr=5e-3)),
    (("build_vae", dict(dataset_size=300, batch_size=20, n_features=40, hidden1=24, hidden2=16, seed=1)), 32, "Adam", dict(lr=1e-3)),
])
def test_two_ranks_walk_the_single_process_trajectory(workload, n_samples, optimizer, opt_kw):
    import torch.multiprocessing as mp
    iters = 12
    ref_losses, ref_params, ref_eval = _single(workload, n_samples, iters, optimizer, opt_kw)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, workload, n_samples, iters, optimizer, opt_kw, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        losses, params, ev, mode, finite, used = got[rank]
        assert finite and "allreduce" in mode and not used, mode
        assert ("+bucket" in mode) == (workload[0] == "build_vae"), mode      # (the amortised path all-reduces the decoder's range early)
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(params, ref_params, rtol=2e-5, atol=2e-6)
        assert abs(ev - ref_eval) <= 2e-5 * abs(ref_eval)
    # the two ranks hold bit-identical parameters (same sums after the all-reduce, same step)
    assert np.array_equal(got[0][1], got[1][1])


@pytest.mark.parametrize("workload,n_samples,optimizer,opt_kw,mode", [
    (("build_readme_ar", dict(T=20)), 600, "SGD", dict(lr=1e-3), "persistent+exchange"),
    (("build_readme_ar", dict(T=20)), 600, "Adam", dict(lr=2e-3), "persistent+exchange"),
    (("build_readme_ar", dict(T=20)), 200, "SGD", dict(lr=1e-3), "persistent+exchange"),      # two sample waves + the draw wave per rank
    (("build_readme_ar", dict(T=40)), 400, "Adam", dict(lr=2e-3), "persistent+exchange"),     # 83 parameters: every thread exchanges its own
    (("build_readme_ar", dict(T=20)), 600, "SGD", dict(lr=1e-3), "graph+allreduce"),
    (("build_logistic_regression", dict(dataset_size=256, batch_size=64, n_features=64, n_classes=10, q_scale=0.05)), 96, "Adam", dict(lr=5e-3),
     "stepwise+allreduce"),
    (("build_vae", dict(dataset_size=300, batch_size=20, n_features=40, hidden1=24, hidden2=16, seed=1)), 32, "Adam", dict(lr=1e-3),
     "stepwise+allreduce"),
])
def test_two_ranks_over_the_one_shot_exchange(workload, n_samples, optimizer, opt_kw, mode):
    """the same trajectories with the opt-in collective BSVI_COLLECTIVE=auto (the default is torch.distributed since round 5): the library's own exchange (bsvi_exchange_*:
    the ranks map each other's regions through HIP IPC, a self-test all-reduce and a vote decide once that it serves) in place
    of the host-staged all-reduce, on all three engines.  On the scalar path the whole loop is ONE launch per rank with the
    exchange inside the kernel's iteration (bsvi_train_persistent_exchange; "persistent+exchange"); with BSVI_LOOP_EXCHANGE=0
    the step sequence — kernel, exchange kernel, finalize — is captured in a HIP graph and replayed ("graph+allreduce")."""
    import torch.multiprocessing as mp
    iters = 12 if mode.startswith("stepwise") else 40
    loop = "0" if mode == "graph+allreduce" else "1"
    ref_losses, ref_params, ref_eval = _single(workload, n_samples, iters, optimizer, opt_kw)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, workload, n_samples, iters, optimizer, opt_kw, q, "auto", loop)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        losses, params, ev, last_mode, finite, used = got[rank]
        assert finite and last_mode == mode, last_mode
        assert used, "the exchange was not chosen (self-test or vote failed): the test would be measuring gloo"
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(params, ref_params, rtol=2e-5, atol=2e-6)
        assert abs(ev - ref_eval) <= 2e-5 * abs(ref_eval)
    assert np.array_equal(got[0][1], got[1][1])


def _abandon_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BSVI_EXCHANGE_TIMEOUT_MS="300", BSVI_COLLECTIVE="auto")
    import time
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brancher_amd import engine, native
    dev = torch.device("cuda", 0)
    first = torch.arange(8, device=dev, dtype=torch.float32) + rank
    engine.allreduce_sums(first)                      # decides for the exchange (self-test + vote), then a good call
    torch.cuda.synchronize()
    ok_first = bool(torch.equal(first, 2 * torch.arange(8, device=dev, dtype=torch.float32) + 1)) and bool(engine._exchanges.get(0))
    dist.barrier()
    block = torch.full((8,), float(rank + 1), device=dev)
    if rank == 1:
        time.sleep(1.5)                               # rank 0 gives up after 0.3 s ...
    engine.allreduce_sums(block)                      # ... and raises the abort word of BOTH regions: the late call is abandoned too
    torch.cuda.synchronize()
    raised = False
    try:
        engine.check_exchange(dev)
    except native.NativeError:
        raised = True
    again = torch.ones(8, device=dev)                 # sticky: nothing after the failure pretends to be a total
    engine.allreduce_sums(again)
    torch.cuda.synchronize()
    out_q.put((rank, ok_first, block.cpu().numpy(), raised, again.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_an_abandoned_exchange_poisons_the_sums_on_every_rank():
    """a rank that gave up waiting used to keep its own partial sums — finite, wrong — while the late peer finished
    the call normally, and only the scalar path's train() ever looked at the status.  Now the rank that gives up raises the
    abort word of every region, an abandoned call leaves NaN in the loss sum and the non-finite count (so bsvi_finalize_step
    skips the optimizer step: the ranks cannot drift apart), the failure is sticky, and `check_exchange` — called at the end
    of every public evaluation / training call of all three engines — raises."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_abandon_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok_first, block, raised, again = got[rank]
        assert ok_first
        assert np.isnan(block[0]) and np.isnan(block[1]), (rank, block)
        assert raised
        assert np.isnan(again[0]) and np.isnan(again[1])


@pytest.mark.parametrize("extra, samples", [
    ([], 300),                                                                  # BASELINE config 1 (the headline), scalar path
    (["--workload", "cfg4", "--samples", "64", "--dataset-size", "4096", "--other-configs", "off"], 64),      # dense-link path
    (["--workload", "cfg5", "--samples", "8", "--other-configs", "off"], 8),                                   # amortised path
])
def test_bench_with_two_ranks_dry_run(extra, samples):
    """`python bench.py --gpus 2` end to end — self-launch, rendezvous, sharded steps, barrier + max-over-ranks timing, ONE
    JSON line from rank 0 — with the two ranks sharing the box's GPU and gloo collectives (a dry run of the flow the driver
    runs on an 8-GPU node with RCCL; not a measurement), for all three engines.  The line says which all-reduce the ranks took."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BSVI_COLLECTIVE")}
    env.update(BSVI_BENCH_BACKEND="gloo", BSVI_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
                          "--spinup-ms", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 40 and line["scaling"] == "weak" and line["all_finite"]
    assert line["config"]["number_samples_global"] == 2 * line["config"]["number_samples_per_gpu"] == 2 * samples
    # (the default collective is torch.distributed's all-reduce — gloo here, RCCL on the driver's node — and the line says so)
    assert "allreduce" in line["config"]["mode"] and line["value"] > 0
    assert line["config"]["collective"].startswith("torch.distributed all_reduce")
    assert "cpu_baseline" not in line                      # rank 0 at N = 1 only
    # ... and the second, guarded phase with BSVI_COLLECTIVE=auto (round 6): both values in the one line — or the reason why not
    alt = {k: v for k, v in line.items() if k.endswith("_alt")}
    alt.update({k: v for k, v in line["config"].items() if "_alt" in k})
    sys.stdout.write("alt phase: %r\n" % (alt,))
    assert ("value_alt" in line and line["value_alt"] > 0 and line["config"].get("collective_alt")) or "collective_alt_error" in line["config"], alt
    assert "collective_alt_error" not in line["config"], alt          # (on this box the phase completes)


def test_bench_falls_back_to_the_host_collective_when_the_exchange_fails_in_use():
    """`bench.py --gpus N` meets a node's topology for the first time at the driver's scaling run: when the one-shot exchange was
    chosen and then abandons a call, every rank sees it (an abandoned call poisons all ranks) and all switch to
    torch.distributed's all-reduce, saying so in the line.  The failure is injected here (both ranks, after a good warm-up)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BSVI_BENCH_BACKEND="gloo", BSVI_BENCH_SHARE_GPU="1", BSVI_COLLECTIVE="auto", BSVI_TEST_HOOKS="1", BSVI_BENCH_INJECT_EXCHANGE_FAILURE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                          "--spinup-ms", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][0]
    assert "collective_fallback" in line["config"] and "allreduce" in line["config"]["mode"] and line["all_finite"]


def test_loop_exchange_on_one_rank_is_the_in_kernel_loop_bit_for_bit(monkeypatch):
    """`bsvi_train_persistent_exchange` with ONE rank: the owners' wave stores its sums into its own region, publishes and
    meets its own sequence number, and reads the same numbers back — so the loss curve and the parameters must be those of
    the plain in-kernel loop, bit for bit, SGD and Adam, with and without the draw wave, across two calls on one exchange
    (the sequence numbers continue), and the exchange must report no abandoned call."""
    sys.path.insert(0, ROOT)
    from brancher_amd import engine, workloads as W
    monkeypatch.setenv("BSVI_LOOP_EXCHANGE", "force")
    monkeypatch.setenv("BSVI_COLLECTIVE", "exchange")       # (opt-in since round 5: the default collective is torch.distributed)
    for n_samples, T in ((300, 20), (128, 20), (200, 40)):      # (T = 40: 83 parameters, exchanged per thread instead of by the owners' wave)
        for optimizer, kw in (("SGD", dict(lr=1e-3)), ("Adam", dict(lr=2e-3))):
            ref = engine.compile_model(W.build_readme_ar(W.native_api(), T=T), None, "pathwise")
            r1, _ = ref.train(25, n_samples, optimizer, seed=3, **kw)
            r2, _ = ref.train(10, n_samples, optimizer, seed=3, **kw)
            assert ref.last_mode == "persistent"
            c = engine.compile_model(W.build_readme_ar(W.native_api(), T=T), None, "pathwise")
            l1, f1 = c.train(25, n_samples, optimizer, seed=3, _force_sharded_path=True, **kw)
            l2, f2 = c.train(10, n_samples, optimizer, seed=3, _force_sharded_path=True, **kw)
            assert c.last_mode == "persistent+exchange", c.last_mode
            assert bool(f1.all()) and bool(f2.all())
            assert torch.equal(l1, r1) and torch.equal(l2, r2), (n_samples, optimizer, (l1 - r1).abs().max().item())
            assert torch.equal(c.params, ref.params)
    engine.check_exchange(torch.device("cuda", 0))
