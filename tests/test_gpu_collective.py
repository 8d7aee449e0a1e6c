"""The exchange of the multi-GPU path behind the C ABI (bsvi_allreduce, bsvi_exchange_*) on the ONE GPU of the test box:
RCCL with a one-rank communicator (two ranks on one device are refused by RCCL), and the one-shot direct-write exchange
between two rank PROCESSES that share the GPU and map each other's regions through HIP IPC — the arrangement of
tests/test_gpu_two_ranks.py.  On a multi-GPU node the same calls run between devices over xGMI (test at the bottom)."""
import os
import socket
import sys

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


def _rccl_worker(port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from brancher_amd import collective
    x = torch.arange(47, dtype=torch.float32, device="cuda") * 0.25
    dist.all_reduce(x.clone())                      # RCCL's first-call set-up
    comm = collective.rccl_comm_ptr()
    ok_ptr = comm is not None
    y = x.clone()
    if ok_ptr:
        collective.rccl_allreduce(y, comm)
        # stream-ordered and capturable: the step sequence of the sharded path replays it from a HIP graph
        z = x.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            collective.rccl_allreduce(z, comm)
        z.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        out_q.put((ok_ptr, bool(torch.equal(y, x)), bool(torch.equal(z, x))))
    else:
        out_q.put((ok_ptr, False, False))
    dist.destroy_process_group()


def test_bsvi_allreduce_runs_on_torchs_rccl_communicator():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    ok_ptr, eager, replayed = q.get(timeout=240)
    p.join(timeout=60)
    assert ok_ptr, "ProcessGroupNCCL gave no communicator pointer"
    assert eager and replayed


def _exchange_worker(rank, world, port, n_calls, skip_call, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BSVI_EXCHANGE_TIMEOUT_MS"] = "300"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brancher_amd import collective
    ex = collective.Exchange(400, device="cuda:0")
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    ok, worst = True, 0.0
    for call in range(n_calls):
        n = [47, 400, 1, 188, 64][call % 5]
        mine = torch.randn(n, generator=g)
        if skip_call is not None and call == skip_call and rank == 1:
            continue                                   # this rank never makes the call: its peer must give up, not hang
        dev = mine.cuda()
        ex.allreduce(dev)
        torch.cuda.synchronize()
        if skip_call is not None:
            continue
        ref = mine.clone()
        dist.all_reduce(ref)                           # the same sum through gloo (two ranks: a + b in either order)
        diff = (dev.cpu() - ref).abs().max().item()
        worst = max(worst, diff)
        ok = ok and torch.equal(dev.cpu(), ref)
    status = ex.status()
    if skip_call is None and n_calls == 0:
        # the self-test of a fresh exchange: the flagged slots AND the tagged 8-byte entries of the in-loop exchange (round 6)
        ok = ex.self_test()
        status = ex.status()
    out_q.put((rank, ok, worst, status))
    dist.barrier()
    ex.close()
    dist.destroy_process_group()


def _run_exchange(n_calls, skip_call):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, n_calls, skip_call, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=240)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
    return got


def test_one_shot_exchange_between_two_rank_processes_sharing_the_gpu():
    got = _run_exchange(40, None)
    for rank in (0, 1):
        ok, worst, status = got[rank]
        assert status == 0
        assert ok, worst                                # bit-identical to the host-side sum, on both ranks


def test_self_test_covers_the_tagged_entries_of_the_in_loop_exchange():
    """`Exchange.self_test` (what `engine._exchange_for` votes on before any rank relies on the exchange): four all-reduces through the
    flagged slots and — round 6 — four through the TAGGED 8-byte entry area that only the in-kernel training loop writes
    (bsvi_exchange_selftest_tagged: the protocol of spec_main.h's spec_xput / spec_xget as a kernel of its own), exact on both ranks"""
    got = _run_exchange(0, None)
    for rank in (0, 1):
        ok, _, status = got[rank]
        assert ok and status == 0, (rank, ok, status)


def test_tagged_self_test_on_one_rank_is_the_identity_and_numbers_its_calls():
    from brancher_amd import collective, native
    import ctypes as C
    ex = collective.Exchange(128, device="cuda:0")
    lib = native.load()
    for call in range(5):
        buf = torch.arange(1, 41, device="cuda:0", dtype=torch.float32) * (call + 1.5)
        want = buf.clone()
        native.check(lib.bsvi_exchange_selftest_tagged(ex.handle, C.c_void_p(buf.data_ptr()), 40, None))
        torch.cuda.synchronize()
        assert torch.equal(buf, want) and ex.status() == 0
    assert lib.bsvi_exchange_selftest_tagged(ex.handle, C.c_void_p(buf.data_ptr()), 65, None) != 0      # one wave: 64 floats at most
    ex.close()


def test_one_shot_exchange_gives_up_instead_of_hanging():
    got = _run_exchange(3, 1)
    assert got[0][2] != 0                              # rank 0 waited for call 2 of rank 1 in vain: abort word set, no hang
    assert got[1][2] == 0 or got[1][2] != 0             # (rank 1 may or may not have met its later calls)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_sharded_training_over_rccl_on_two_gpus():
    """the first multi-GPU run of the graph-captured step sequence with a two-rank RCCL all-reduce inside (skips on the
    one-GPU test box): `bench.py --gpus 2` end to end"""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["all_finite"] and line["config"]["mode"] in ("graph+allreduce", "stepwise+allreduce")
