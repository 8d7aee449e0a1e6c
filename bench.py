#!/usr/bin/env python3
"""
bench.py — ELBO iterations/sec of the fused SVI engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload cfg1|cfg2|cfg3] [--mode auto|persistent|stepwise]

A "step" is one complete SVI iteration of `brancher/inference.py:95-108`: draw number_samples
reparameterised posterior samples (in-kernel Philox), evaluate log p + entropy over the model
graph, reverse sweep, (all-reduce over GPUs), finite check, optimizer step, loss log.
Default workload = BASELINE config the metric is quoted on: the README T=20 autoregressive
model at number_samples=300, SGD lr=1e-3 (`README.md:25-75`).

Multi-GPU (launched by torch.distributed.run, one rank per GPU): weak scaling over the
Monte-Carlo sample axis — every GPU evaluates `number_samples` samples of the same
iteration and ONE all-reduce (RCCL) of 4+P floats joins them.  `value` is whole-job
throughput in 300-sample ELBO iterations per second:  (global samples per step / 300) * steps / s.

One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (builder, kwargs, number_samples per GPU, optimizer, opt kwargs, description)
    "cfg1": ("build_readme_ar", dict(T=20), 300, "SGD", dict(lr=1e-3),
             "README AR state-space T=20, number_samples=300, SGD lr=1e-3 (BASELINE config 1)"),
    "cfg2": ("build_beta_binomial", dict(n_obs=30), 4096, "SGD", dict(lr=0.1),
             "beta_binomial.py Beta posterior, 30 observations, number_samples=4096, SGD lr=0.1 (BASELINE config 2)"),
    "cfg3": ("build_readme_ar", dict(T=200), 1024, "SGD", dict(lr=1e-4),
             "README AR state-space T=200, number_samples=8192 sharded as 1024 per GPU (BASELINE config 3)"),
    "cfg4": ("build_logistic_regression", dict(dataset_size=60000, batch_size=512, n_features=784, n_classes=10),
             1024, "Adam", dict(lr=5e-3),
             "Bayesian multinomial logistic regression, dense matmul link 10x784, minibatch 512 of 60000 synthetic "
             "rows, number_samples=1024, Adam lr=5e-3 (BASELINE config 4)"),
    "cfg5": ("build_vae", dict(dataset_size=60000, batch_size=100, n_features=784, latent_size=2, hidden1=512, hidden2=256),
             256, "Adam", dict(lr=1e-3),
             "VAE_playground.py MLP VAE 784-256-512-(2,2) / 2-512-256-784, Binomial(1, logits) likelihood, every sample "
             "draws its own minibatch of 100 of 60000 synthetic binary rows, number_samples=2048 sharded as 256 per GPU "
             "(25600 rows per GPU), Adam lr=1e-3 (BASELINE config 5)"),
    "cfg1_big": ("build_readme_ar", dict(T=20), 262144, "SGD", dict(lr=1e-3),
                 "README AR T=20 at number_samples=262144 (throughput regime of the same kernel)"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_DIR = os.path.join(ROOT, "profiles", "r2")


def pmc_traffic_bytes(csv_name, kernel_substrings, double_fetch=False, column="mean_KB_per_dispatch"):
    """HBM bytes per launch of the named kernel(s) from the committed rocprofv3 PMC summary (FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc passes by tools/pmc_hbm.sh, KB per dispatch).  double_fetch applies
    the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies the 128-B requests of 16-B-per-lane
    streaming reads at 64 B).  None when the summary is not there."""
    import csv
    path = os.path.join(PROFILE_DIR, csv_name)
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r1", csv_name)      # (the dense / amortised kernels were last profiled in round 1)
    if not os.path.exists(path):
        return None
    total = 0.0
    hit = False
    for row in csv.DictReader(open(path)):
        if any(k in row["kernel"] for k in kernel_substrings):
            kb = float(row[column])
            total += kb * 1024.0 * (2.0 if double_fetch and row["counter"] == "FETCH_SIZE" else 1.0)
            hit = True
    return total if hit else None
def pmc_dispatches(csv_name, kernel_substring):
    """number of dispatches of a kernel in a committed PMC summary (None when absent)"""
    import csv
    for d in (PROFILE_DIR, os.path.join(ROOT, "profiles", "r1")):
        path = os.path.join(d, csv_name)
        if os.path.exists(path):
            for row in csv.DictReader(open(path)):
                if kernel_substring in row["kernel"]:
                    return int(row["dispatches"])
            return None
    return None


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA = f32 vector peak


def dense_flops_per_iteration(program, n_local):
    """SURVEY §8d cfg 4: forward 2*N*C*P*B for the logits GEMM and the same again for the
    weight-gradient GEMM."""
    return 2 * 2.0 * n_local * program.n_classes * program.n_features * program.batch_size


def algorithmic_bytes_per_iteration(program, n_local):
    """N*L*4 noise bytes (the eps the estimator consumes; generated in registers here)
    + parameters read + gradients written."""
    return n_local * program.n_noise * 4 + 2 * program.n_params * 4


def amort_flops_per_iteration(program, n_local):
    """SURVEY §8d cfg 5: every Linear layer is three GEMMs of 2*R*n_in*n_out flops over the R = N*B rows of the
    iteration — forward, weight gradient, input gradient (the last not for layers reading the data rows)."""
    rows = n_local * program.batch_size
    flops = 0.0
    for net, input_grad in ((program.enc_layers, False), (program.dec_layers, True)):
        for l in net:
            flops += 2.0 * rows * l.n_in * l.n_out * (3 if (l.in_value != 0 or input_grad) else 2)
    return flops


def cpu_baseline_vae(kwargs, optimizer, opt_kwargs, budget_s=12.0, n_cpu=8):
    """oracle/vae_oracle.py (PyTorch-CPU autograd through the same torch modules the reference calls) on a bounded
    sample: number_samples=8 (800 rows) per iteration, up to 16 threads."""
    import numpy as np
    import torch
    from brancher_amd import workloads as W
    from oracle.vae_oracle import VaeOracle
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    oracle = VaeOracle(getattr(W, "build_vae")(W.native_api(), **kwargs))
    rng = np.random.RandomState(0)
    DS, B = kwargs["dataset_size"], kwargs["batch_size"]

    def draws(k):
        rows = [np.stack([rng.choice(DS, B, replace=False) for _ in range(n_cpu)]) for _ in range(k)]
        eps = [rng.randn(n_cpu, B, oracle.Dz).astype(np.float32) for _ in range(k)]
        return rows, eps

    oracle.train(1, *draws(1), optimizer, **opt_kwargs)
    iters, spent = 0, 0.0
    while spent < budget_s and iters < 400:
        rows, eps = draws(4)
        t0 = time.perf_counter()
        oracle.train(4, rows, eps, optimizer, **opt_kwargs)
        spent += time.perf_counter() - t0
        iters += 4
    return dict(value=iters / spent * (n_cpu / 300.0), unit="it/s", cores=cores, kind="port",
                sample="%d iterations of the same workload at number_samples=%d (%d rows) in %.1f s, oracle/vae_oracle.py "
                       "on PyTorch-CPU, %d thread(s)" % (iters, n_cpu, n_cpu * B, spent, cores),
                iters_per_sec=iters / spent, number_samples=n_cpu)


def cpu_baseline(builder, kwargs, n_samples, optimizer, opt_kwargs, dense=False, budget_s=12.0, max_iters=400):
    """The oracle (PyTorch-CPU restatement of the reference loop, kind 'port') timed on the host.
    Scalar graphs are dispatch-bound (more threads are slower): 1 thread, batches of 5
    iterations at the full number_samples.  One dense-link iteration at number_samples=1024 takes the oracle
    about a minute, so the bounded sample there is number_samples=64 on up to 16 threads; `value` is in the
    metric's unit either way (300-sample-equivalent iterations per second)."""
    import torch
    from brancher_amd import workloads as W
    from oracle.svi_oracle import Oracle
    cores = min(os.cpu_count() or 1, 16) if dense else 1
    torch.set_num_threads(cores)
    batch = 1 if dense else 5
    n_cpu = min(n_samples, 64) if dense else n_samples
    oracle = Oracle(getattr(W, builder)(W.native_api(), **kwargs))
    torch.manual_seed(0)
    oracle.train(1 if dense else 2, n_cpu, optimizer, "pathwise", None, **opt_kwargs)
    t0 = time.perf_counter()
    iters = 0
    while iters < max_iters:
        oracle.train(batch, n_cpu, optimizer, "pathwise", None, **opt_kwargs)
        iters += batch
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=iters / dt * (n_cpu / 300.0), unit="it/s", cores=cores, kind="port",
                sample="%d iterations of the same workload at number_samples=%d in %.1f s, oracle/svi_oracle.py "
                       "on PyTorch-CPU, %d thread(s)" % (iters, n_cpu, dt, cores),
                iters_per_sec=iters / dt, number_samples=n_cpu)


def recorded_reference_timings(workload, n_samples, optimizer):
    """The REAL reference's iteration rate on this workload, recorded once in the build container by
    oracle/time_reference.py (the reference cannot travel to the GPU box) and reported with the hardware it was taken
    on: a fixture, not a live measurement -- the live CPU number is the oracle port above."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "reference_cpu_timings.json")
    if workload != "cfg1" or n_samples != 300 or optimizer != "SGD" or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    best = max(rec["runs"], key=lambda r: r["iters_per_sec"])
    return dict(kind="reference", live=False, value=best["iters_per_sec"], unit="it/s", cores=best["threads"],
                hardware=rec["hardware"], torch=rec["torch"], runs=rec["runs"], sample="%d iterations, %s"
                % (best["iterations_timed"], rec["workload"]))


def self_launch(n_gpus, argv=None, port=None):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child (one rank per GPU, rendezvous on 127.0.0.1) and pass its output and exit code on."""
    import socket
    import subprocess
    if port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg1", choices=sorted(WORKLOADS))
    ap.add_argument("--dataset-size", type=int, default=0, help="cfg4: override the synthetic dataset size")
    ap.add_argument("--mode", default="auto", choices=["auto", "persistent", "stepwise"])
    ap.add_argument("--samples", type=int, default=0, help="override number_samples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed iterations of the timed path after the W warm-up steps, to let the clocks ramp (0: off)")
    ap.add_argument("--estimator", default="pathwise", choices=["pathwise", "blackbox", "taylor1"],
                    help="gradient estimator (BASELINE config 5 names both Pathwise and BlackBox)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU): every rank contributes its rank to one all-reduce and rank 0 "
                         "prints what arrived; exercises the launcher path of --gpus N on a CPU-only box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain `python bench.py --gpus N`: become the launcher.  The ranks are CHILD processes started
        # before this process has touched the GPU (nothing here has initialised HIP yet); their output is relayed.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        seen = torch.tensor([float(rank), 1.0])
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps(dict(launch_check=True, world=world, rank_sum=seen[0].item(), ranks=seen[1].item(),
                                  local_rank=local_rank)))
        dist.destroy_process_group()
        return
    # BSVI_BENCH_SHARE_GPU=1 with BSVI_BENCH_BACKEND=gloo: dry run of the N > 1 flow on a box with fewer GPUs than ranks
    # (the ranks share devices, collectives go through the host) — tests/test_gpu_two_ranks.py; never a measurement
    backend = os.environ.get("BSVI_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if os.environ.get("BSVI_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            os.environ["BSVI_GRAPH"] = "0"          # host-staged collectives cannot be captured into a HIP graph
            dist.init_process_group(backend)

    from brancher_amd import config, engine, workloads as W
    config.set_device("cuda:%d" % device_index)
    builder, kwargs, n_per_gpu, optimizer, opt_kwargs, desc = WORKLOADS[args.workload]
    if args.samples:
        n_per_gpu = args.samples
    if args.dataset_size and "dataset_size" in kwargs:
        kwargs = dict(kwargs, dataset_size=args.dataset_size)
    n_global = n_per_gpu * world
    model = getattr(W, builder)(W.native_api(), **kwargs)
    compiled = engine.compile_model(model, None, args.estimator)
    program = compiled.program
    allow_persistent = args.mode != "stepwise"
    if args.mode == "auto" and getattr(compiled, "prefers_stepwise", None) and compiled.prefers_stepwise(n_global):
        allow_persistent = False      # 4+ program shares: launches per iteration beat the persistent trainer
    if args.mode == "persistent" and not (world == 1 and hasattr(compiled, "native") and compiled.native.persistent_supported(n_per_gpu)):
        raise SystemExit("persistent mode needs one GPU and a sample count that fits one workgroup")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed): W steps through exactly the path that is timed (this is also where hiprtc compiles the
    #      program-specialised kernel), then — still untimed — the same path for --spinup-ms so that a short timed region
    #      (the driver runs K = 20) is not measured on a GPU whose clocks have not ramped yet
    compiled.train(max(args.warmup, 1), n_global, optimizer, seed=0, allow_persistent=allow_persistent, **opt_kwargs)
    barrier()
    # no garbage collection from here to the end of the timed region: a collection inside a 200 us region would be a
    # tenth of it, and a pause between the spin-up and the region would let the clocks drop again
    import gc
    gc.collect()
    gc.disable()
    spun = 0
    if args.spinup_ms > 0:
        chunk = max(args.steps, 200)
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            compiled.train(chunk, n_global, optimizer, seed=0, allow_persistent=allow_persistent, **opt_kwargs)
            torch.cuda.synchronize()
            spun += chunk
    barrier()

    # ---- timed region: exactly K steps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    losses, finite = compiled.train(args.steps, n_global, optimizer, seed=0, allow_persistent=allow_persistent,
                                    **opt_kwargs)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    dev_ms = ev0.elapsed_time(ev1)
    mode = compiled.last_mode

    t = torch.tensor([dt, dev_ms], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dev_ms = float(t[0]), float(t[1])
    ok = bool(torch.isfinite(losses).all()) and bool(finite.all())

    if rank == 0:
        iters_per_sec = args.steps / dt
        value = iters_per_sec * (n_global / 300.0)
        dense = hasattr(program, "n_classes")
        amort = hasattr(program, "enc_layers")
        geom = dict(kind="dense") if dense else dict(kind="amortized") if amort else compiled.native.geometry(n_per_gpu)
        spec = None
        if not dense and not amort:
            spec = compiled.native.engine(n_per_gpu, 2 if mode == "persistent" else 1 if mode == "stepwise" else 0)
            spec = spec if spec["engine"] == "specialised" else None
        # roofline of the dominant kernel (the fused ELBO kernel; in persistent mode one launch
        # covers all K iterations).  Launch duration from HIP events on the launch stream.
        alg_bytes_iter = algorithmic_bytes_per_iteration(program, n_per_gpu * (program.batch_size if amort else 1))
        if spec is not None:
            # the program-specialised kernel: ONE launch per bsvi_* call — the whole loop in persistent mode
            units_per_launch = args.steps if mode == "persistent" else 1
            launch_ms = dev_ms / (1 if mode == "persistent" else args.steps)
            kernel = "bsvi_spec_kernel (straight-line HIP generated from the model program, hiprtc)"
            geom = dict(engine="specialised", n_blocks=spec["n_blocks"], n_threads=spec["n_threads"],
                        lds_bytes=spec["lds_bytes"], storage="registers")
        elif mode == "persistent":
            launch_ms, launches, units_per_launch = dev_ms, 1, args.steps
            # five or more waves run as one wave per workgroup (persistent_multi_kernel)
            multi = (n_per_gpu + 63) // 64 >= 5 and os.environ.get("BSVI_PERSISTENT_MULTI", "1") != "0"
            kernel = "bsvi::persistent_%skernel<%s>" % ("multi_" if multi else "", geom.get("storage", "") or "lds+lane_acc")
            if multi:
                shares = int(compiled.lib.bsvi_persistent_split_shares(compiled.native.handle, n_per_gpu))
                if not getattr(program, "shares", {}).get(shares):
                    shares = 1
                geom = dict(geom, n_blocks=(n_per_gpu + 63) // 64 * shares, n_waves=1, storage="lds+lane_acc",
                            program_shares=shares,
                            note="one wave per workgroup; workgroup w runs share w %% %d of the model's log-prob records "
                                 "on sample wave w / %d; one exchange of partial sums per iteration" % (shares, shares))
        else:
            launch_ms, launches, units_per_launch = dev_ms / args.steps, args.steps, 1
            kernel = "bsvi::elbo_kernel<%s>" % geom.get("storage", "dense")
            V = getattr(getattr(compiled, "native", None), "_elbo_shares_set", 0)
            if not dense and not amort and V >= 2 and os.environ.get("BSVI_ELBO_SHARES", "1") != "0":
                waves = (n_per_gpu + 63) // 64
                if geom.get("storage") == "lds+lane_acc" and geom.get("lanes_per_wave") == 64 and waves * V <= 256:
                    geom = dict(geom, n_blocks=waves, n_waves=1)        # one wave per workgroup (share_geometry)
                geom = dict(geom, n_blocks=geom["n_blocks"] * V, program_shares=V,
                            note="workgroup b runs share b %% %d of the model's log-prob records on sample group b / %d; "
                                 "reduce_kernel adds the rows of partial sums" % (V, V))
        achieved = alg_bytes_iter * units_per_launch / (launch_ms * 1e-3) / 1e9
        if spec is not None:
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/pmc_hbm.sh), committed per round:
            # the timed launch is the larger of the two launches in the summary (the other is the warm-up)
            if args.workload == "cfg1" and not args.samples and args.steps == 20000 and mode == "persistent":
                traffic = pmc_traffic_bytes("cfg1_spec_pmc_hbm_traffic.csv", ["bsvi_spec_kernel"], column="max_KB")
            elif args.workload == "cfg1" and not args.samples and args.steps == 20 and mode == "persistent":
                # (the driver's command line: every launch of that profile run is one 20-iteration launch)
                traffic = pmc_traffic_bytes("cfg1_spec_k20_pmc_hbm_traffic.csv", ["bsvi_spec_kernel"])
            elif args.workload == "cfg1" and not args.samples and mode == "stepwise":
                traffic = pmc_traffic_bytes("cfg1_spec_stepwise_pmc_hbm_traffic.csv", ["bsvi_spec_kernel"])
            else:
                traffic = None
        elif mode == "persistent":
            traffic = None
        else:
            traffic = pmc_traffic_bytes("pmc_hbm_traffic.csv", ["elbo_kernel"]) \
                if args.workload == "cfg1" and not args.samples else None
        roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=traffic, kernel=kernel, algorithmic_bytes_per_iteration=alg_bytes_iter,
                        iterations_per_launch=units_per_launch, launch_ms=launch_ms,
                        note="latency / issue-bound workload (SURVEY §8d cfg 1): %.2f us per iteration; HBM is not its "
                             "roof — the noise is generated in registers, a launch fetches ~25 KB — what bounds it is the "
                             "serial instruction stream of one wave per sample group (cfg 1: 1 335 VALU instructions per "
                             "wave and iteration, profiles/r2/pmc_sq_loop.csv) and two workgroup barriers per iteration; "
                             "a launch-per-step design pays ~5-10 us of launch latency per iteration on top"
                             % (dev_ms * 1e3 / args.steps))
        if dense:
            # the whole iteration (8 launches) is timed; the two MFMA GEMMs are >90 % of it (profiles/)
            flops = dense_flops_per_iteration(program, n_per_gpu)
            tf = flops / (dev_ms * 1e-3 / args.steps) / 1e12
            traffic = pmc_traffic_bytes("cfg4_pmc_hbm_traffic.csv", ["dense_forward", "dense_backward"],
                                        double_fetch=True) if args.workload == "cfg4" and not args.samples else None
            roofline = dict(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=tf / MFMA_F32_PEAK_TFLOPS, traffic=traffic,
                            kernel="bsvi::dense_forward<10> + bsvi::dense_backward",
                            algorithmic_flops_per_iteration=flops, launch_ms=dev_ms / args.steps,
                            note="f32-input MFMA (v_mfma_f32_16x16x4_f32); achieved = GEMM flops of one iteration / "
                                 "duration of the whole iteration")
        if amort:
            # the whole iteration (~35 launches) is timed; the 20 MFMA GEMMs carry the flops
            flops = amort_flops_per_iteration(program, n_per_gpu)
            tf = flops / (dev_ms * 1e-3 / args.steps) / 1e12
            # HBM bytes of ALL launches of one iteration (tools/pmc_hbm.sh over a short run of this workload;
            # FETCH_SIZE doubled per the guide's gfx950 rule for 16-byte-per-lane reads)
            traffic = pmc_traffic_bytes("cfg5_pmc_hbm_traffic.csv", ["bsvi_amort_impl"], double_fetch=True,
                                        column="total_KB") if args.workload == "cfg5" and not args.samples else None
            iterations_profiled = pmc_dispatches("cfg5_pmc_hbm_traffic.csv", "amort_rows")   # one launch per iteration
            traffic = traffic / iterations_profiled if traffic and iterations_profiled else None
            roofline = dict(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=tf / MFMA_F32_PEAK_TFLOPS, traffic=traffic,
                            kernel="bsvi_amort_impl::gemm_kernel<0|1|2>",
                            algorithmic_flops_per_iteration=flops, launch_ms=dev_ms / args.steps,
                            rows_per_iteration=n_per_gpu * program.batch_size,
                            note="f32-input MFMA (v_mfma_f32_32x32x2_f32); achieved = GEMM flops of one iteration "
                                 "(forward + weight gradient + input gradient of every Linear layer) / duration of "
                                 "the whole iteration")
        line = dict(metric="ELBO iters/sec at num_samples=%d per GPU (300-sample-equivalent iterations, whole job)"
                           % n_per_gpu,
                    value=value, unit="it/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=dt * 1e3 / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="f32", data="synthetic",
                    config=dict(workload=desc, number_samples_per_gpu=n_per_gpu, number_samples_global=n_global,
                                optimizer=optimizer, **{k: v for k, v in opt_kwargs.items()},
                                estimator=args.estimator, mode=mode, parallelism="sample-shard x%d" % world,
                                grid=geom, untimed_spinup_iterations=spun),
                    iters_per_sec=iters_per_sec, samples_per_sec=iters_per_sec * n_global,
                    device_ms_per_step=dev_ms / args.steps, all_finite=ok,
                    final_loss=float(losses[-1].item()), roofline=roofline)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_vae(kwargs, optimizer, opt_kwargs) if amort else \
                cpu_baseline(builder, kwargs, n_per_gpu, optimizer, opt_kwargs, dense=dense)
            recorded = recorded_reference_timings(args.workload, n_per_gpu, optimizer)
            if recorded is not None:
                line["cpu_baseline"]["reference"] = recorded
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
