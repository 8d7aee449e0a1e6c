#!/usr/bin/env python3
"""
bench.py — ELBO iterations/sec of the fused SVI engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload cfg1|...|cfg5] [--mode auto|persistent|stepwise]

A "step" is one complete SVI iteration of `brancher/inference.py:95-108`: draw number_samples
reparameterised posterior samples (in-kernel Philox), evaluate log p + entropy over the model
graph, reverse sweep, (all-reduce over GPUs), finite check, optimizer step, loss log.
Default workload = BASELINE config the metric is quoted on: the README T=20 autoregressive
model at number_samples=300, SGD lr=1e-3 (`README.md:25-75`).

Multi-GPU (launched by torch.distributed.run, one rank per GPU): weak scaling over the
Monte-Carlo sample axis — every GPU evaluates `number_samples` samples of the same
iteration and ONE all-reduce (RCCL) of 4+P floats joins them.  `value` is whole-job
throughput in 300-sample ELBO iterations per second:  (global samples per step / 300) * steps / s.

One JSON line is printed by rank 0.  On one GPU with the default workload the same line also carries BASELINE
configs 2-5 (`other_configs`: each timed for 50 iterations in this process after the headline's timed region, with its
own roofline object), and `roofline.traffic` is measured live (rocprofv3 --pmc passes of this script as child
processes before the GPU is touched here; see run_traffic_probe).
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
    "cfg4": ("build_logistic_regression", dict(dataset_size=60000, batch_size=512, n_features=784, n_classes=10, pixels="uint8",
                                               q_scale=0.01),
             1024, "Adam", dict(lr=5e-3),
             "Bayesian multinomial logistic regression, dense matmul link 10x784, minibatch 512 of 60000 synthetic "
             "rows of uint8-valued pixels (SURVEY 8d; examples/MNIST_logistic_regression.py feeds raw pixel counts), "
             "number_samples=1024, Adam lr=5e-3 (BASELINE config 4)"),
    "cfg4_unit": ("build_logistic_regression", dict(dataset_size=60000, batch_size=512, n_features=784, n_classes=10),
                  1024, "Adam", dict(lr=5e-3),
                  "config 4 with features in [0, 1] (not bf16 numbers: the f32-input MFMA kernels serve the products)"),
    "cfg5": ("build_vae", dict(dataset_size=60000, batch_size=100, n_features=784, latent_size=2, hidden1=512, hidden2=256),
             256, "Adam", dict(lr=1e-3),
             "VAE_playground.py MLP VAE 784-256-512-(2,2) / 2-512-256-784, Binomial(1, logits) likelihood, every sample "
             "draws its own minibatch of 100 of 60000 synthetic binary rows, number_samples=2048 sharded as 256 per GPU "
             "(25600 rows per GPU), Adam lr=1e-3 (BASELINE config 5)"),
    "bnn": ("build_bayesian_neural_network", dict(dataset_size=60000, batch_size=30, n_features=784, n_hidden=20, n_classes=10,
                                                  q_scale1=4e-4, q_loc_scale=1.0),
            50, "Adam", dict(lr=5e-3),
            "Bayesian neural network 784-20-10 (tanh), minibatch 30 of 60000 synthetic rows, number_samples=50, Adam lr=5e-3 "
            "(tests/test_MNIST_bayesian_neural_network.py:20-60 of the reference: its example's size)"),
    "bnn_cfg4scale": ("build_bayesian_neural_network", dict(dataset_size=60000, batch_size=512, n_features=784, n_hidden=20, n_classes=10,
                                                            q_scale1=4e-4, q_loc_scale=1.0),
                      1024, "Adam", dict(lr=5e-3),
                      "the same Bayesian neural network at BASELINE config 4's scale: minibatch 512, number_samples=1024"),
    "cfg1_big": ("build_readme_ar", dict(T=20), 262144, "SGD", dict(lr=1e-3),
                 "README AR T=20 at number_samples=262144 (throughput regime of the same kernel)"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
OTHER_CONFIGS = ("cfg2", "cfg3", "cfg4", "cfg5", "bnn", "bnn_cfg4scale")     # timed after the headline in the same process (N = 1)
# BASELINE.json config 5 names "BlackBox + Pathwise estimators": the two MFMA configs once more under BlackBox (no traffic pass of their own)
OTHER_BLACKBOX = ("cfg4", "cfg5")
# (the same untimed spin-up as the headline: the clocks of an idle MI355X take longer than 100 ms to ramp — cfg 2, a pure
#  latency chain, measured 49.9 / 34.7 / 28.8 / 27.1 us per iteration after 0 / 100 / 300 / 1000 ms; every config's `cold_start`
#  is in the line beside its hot figure)
OTHER_STEPS, OTHER_WARMUP, OTHER_SPINUP_MS = 50, 5, 300.0


# ---- HBM traffic, measured live ----------------------------------------------------------------------------------
# `roofline.traffic` is the HBM bytes the PMC counters saw for the SAME launches this run times: before this process
# touches the GPU it runs itself twice as a child under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and
# WRITE_SIZE in separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes; the child is `--traffic-probe`:
# the same workloads, the same step counts, no CPU baseline) and reads the per-dispatch counter values back.  No profiler
# on the box, or a failed pass -> traffic is null and `traffic_note` says why; nothing is looked up in committed files.
def run_traffic_probe(workloads, steps_of, warmup_of, estimator, mode, samples, timeout_s=420):
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found on this box"
    rows = []
    spec = ",".join("%s:%d:%d" % (w, steps_of[w], warmup_of[w]) for w in workloads)
    # (a third pass counts the vector instructions of every launch: roofline.issue of the latency-bound configs)
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        out_dir = tempfile.mkdtemp(prefix="bsvi_pmc_%s_" % counter, dir="/tmp")
        cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--",
               sys.executable, os.path.abspath(__file__), "--traffic-probe", spec, "--estimator", estimator, "--mode", mode]
        if samples:
            cmd += ["--samples", str(samples)]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                                 stderr=subprocess.STDOUT, timeout=timeout_s, text=True)
        except Exception as err:      # noqa: BLE001  (timeout, exec failure: the bench goes on without the counters)
            shutil.rmtree(out_dir, ignore_errors=True)
            if counter == "SQ_INSTS_VALU":
                break                 # (the traffic passes stand on their own)
            return None, "PMC pass %s did not finish: %s" % (counter, err)
        files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
        if res.returncode != 0 or not files:
            shutil.rmtree(out_dir, ignore_errors=True)
            if counter == "SQ_INSTS_VALU":
                break
            return None, "PMC pass %s failed (exit %d): %s" % (counter, res.returncode, (res.stdout or "")[-300:])
        per = {}
        for f in files:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter or not ("bsvi" in r["Kernel_Name"] or "bnn_" in r["Kernel_Name"]):
                    continue          # (the library's kernels; bnn_*: so that a config's span of the probe ends where the network's begins)
                key = (int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]))
                per[key] = per.get(key, 0.0) + float(r["Counter_Value"])          # (one row per XCD / instance)
        for (dispatch, kernel, grid), kb in sorted(per.items()):
            rows.append(dict(counter=counter, dispatch=dispatch, kernel=kernel, grid=grid, bytes=kb * 1024.0, value=kb))
        shutil.rmtree(out_dir, ignore_errors=True)
    return rows, None


def traffic_of(rows, kernel_substrings, grid=None, last_only=False, per=1, double_fetch=False):
    """HBM bytes (FETCH_SIZE + WRITE_SIZE) of the probe's dispatches whose kernel name contains one of the substrings (and
    whose grid matches): the LAST such dispatch of each counter (`last_only`: the timed launch of an in-kernel loop), else
    their sum divided by `per` (iterations run by the probe).  double_fetch: the gfx950 rule of MI355X_MICROARCH.md for
    16-byte-per-lane streaming reads (FETCH_SIZE tallies their 128-B requests at 64 B)."""
    if not rows:
        return None
    total, hit = 0.0, False
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        sel = [r for r in rows if r["counter"] == counter and any(k in r["kernel"] for k in kernel_substrings)
               and (grid is None or r["grid"] == grid)]
        if not sel:
            continue
        hit = True
        scale = 2.0 if (double_fetch and counter == "FETCH_SIZE") else 1.0
        total += scale * (sel[-1]["bytes"] if last_only else sum(r["bytes"] for r in sel) / max(per, 1))
    return total if hit else None


def issue_roofline(rows, kernel_substrings, grid, n_samples, iterations, us_per_iteration, last_only):
    """roofline.issue of a latency-bound config: the vector instructions ONE wave issues per iteration (SQ_INSTS_VALU of
    the launch / its waves / its iterations, from this run's own counter pass) x 4 cycles of issue each
    (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost') x the waves that share the busiest SIMD, at the 2.4 GHz peak
    clock, against the measured iteration.  The launch is ONE workgroup on one CU's four SIMDs: `grid` / 64 waves — the
    sample waves and, in the in-kernel loop, the draw waves beside them (round 5: eight waves at 300 samples)."""
    sel = [r for r in (rows or []) if r["counter"] == "SQ_INSTS_VALU" and any(k in r["kernel"] for k in kernel_substrings)
           and (grid is None or r["grid"] == grid)]
    if not sel:
        return None
    insts = sel[-1]["value"] if last_only else sum(r["value"] for r in sel) / len(sel)
    waves = max((n_samples + 63) // 64, int(grid) // 64 if grid else 0)
    per_wave_iter = insts / waves / max(iterations, 1)
    busiest = (waves + 3) // 4
    issue_us = per_wave_iter * 4.0 * busiest / 2400.0
    return dict(bound="valu-issue of the busiest SIMD", valu_per_wave_iteration=per_wave_iter, waves=waves,
                waves_on_busiest_simd=busiest, cycles_per_instruction=4, clock_ghz=2.4, issue_us_per_iteration=issue_us,
                measured_us_per_iteration=us_per_iteration, frac=issue_us / us_per_iteration,
                note="one workgroup of %d waves on ONE of 256 CUs: the iteration cannot be shorter than the instruction issue "
                     "of the SIMD that hosts %d of them; the rest of the iteration is barriers, LDS exchange and dependent "
                     "latency (DESIGN.md 4.7)" % (waves, busiest))


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA = f32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (16x the f32-input rate)
# a product whose data operand is exactly bf16 runs as THREE bf16 MFMAs on the exact pieces of its f32 operand:
MFMA_EXACT_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 3.0


def dense_flops_per_iteration(program, n_local):
    """SURVEY §8d cfg 4: forward 2*N*C*P*B for the logits GEMM and the same again for the
    weight-gradient GEMM."""
    return 2 * 2.0 * n_local * program.n_classes * program.n_features * program.batch_size


def algorithmic_bytes_per_iteration(program, n_local):
    """N*L*4 noise bytes (the eps the estimator consumes; generated in registers here)
    + parameters read + gradients written."""
    return n_local * program.n_noise * 4 + 2 * program.n_params * 4


def dense_bytes_per_iteration(program, n_local):
    """cfg 4, stated both ways: SURVEY 8d's figure — the [N, C, P] noise the estimator consumes (N*C*P*4: never stored
    here, drawn in registers twice) + the minibatch [B, P] and its labels + parameters read and gradients written — and what THIS
    design must move: the minibatch as bf16 (its gather source rows are f32), labels, parameters, gradients."""
    noise = n_local * program.n_classes * program.n_features * 4
    batch = program.batch_size * (program.n_features + 1) * 4
    par = 2 * program.n_params * 4
    return dict(survey=noise + batch + par, must_move=batch + par)


def amort_bytes_per_iteration(program, n_local):
    """cfg 5, layer by layer as this design (and the reference's PyTorch graph) runs it — every Linear layer writes its activation
    and reads its input in the forward pass, reads dY twice (input and weight gradient), its input and the activation it
    differentiates, and writes dX in the backward pass; the likelihood reads the logits and the data rows and writes dlogits; the
    data rows are gathered three times (first layer, likelihood, first layer's weight gradient); parameters read, gradients written.
    What a fused multi-layer kernel would NOT have to move (activations that stay on chip) is in here: it is the byte count of the
    layer-by-layer formulation, the denominator for `traffic`."""
    rows = n_local * program.batch_size
    P = program.n_features
    total = 3.0 * rows * P * 4                                   # the gathered data rows, three times
    for net, input_grad in ((program.enc_layers, False), (program.dec_layers, True)):
        for l in net:
            total += rows * (l.n_in + l.n_out) * 4.0             # forward: read x, write y
            total += rows * (l.n_out + l.n_in) * 4.0             # weight gradient: read dY and x
            if l.in_value != 0 or input_grad:
                total += rows * (l.n_out + 2 * l.n_in) * 4.0     # input gradient: read dY and the activation, write dX
    total += rows * P * 2 * 4.0                                  # the likelihood: read the logits, write their gradient
    total += 2 * program.n_params * 4.0
    return total


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
    """The oracle (PyTorch-CPU restatement of the reference loop, kind 'port') timed on the host, LIVE, at ONE thread and at ALL host
    cores ("all host cores with the core count printed"): `value` / `cores` are the faster of the two legs, `runs`
    holds both.  Scalar graphs are dispatch-bound (more threads are slower): batches of 5 iterations at the full
    number_samples.  One dense-link iteration at number_samples=1024 takes the oracle about a minute, so the bounded sample there is
    number_samples=64; `value` is in the metric's unit either way (300-sample-equivalent iterations per second)."""
    import torch
    from brancher_amd import workloads as W
    from oracle.svi_oracle import Oracle
    all_cores = min(os.cpu_count() or 1, 16)
    batch = 1 if dense else 5
    n_cpu = min(n_samples, 64) if dense else n_samples
    oracle = Oracle(getattr(W, builder)(W.native_api(), **kwargs))
    runs = []
    legs = [1] if all_cores == 1 else [1, all_cores]
    for cores in legs:
        torch.set_num_threads(cores)
        torch.manual_seed(0)
        oracle.train(1 if dense else 2, n_cpu, optimizer, "pathwise", None, **opt_kwargs)
        t0 = time.perf_counter()
        iters = 0
        while iters < max_iters:
            oracle.train(batch, n_cpu, optimizer, "pathwise", None, **opt_kwargs)
            iters += batch
            if time.perf_counter() - t0 > budget_s / len(legs):
                break
        dt = time.perf_counter() - t0
        runs.append(dict(cores=cores, value=iters / dt * (n_cpu / 300.0), iters_per_sec=iters / dt, iterations=iters, seconds=dt))
    torch.set_num_threads(1)
    best = max(runs, key=lambda r: r["value"])
    return dict(value=best["value"], unit="it/s", cores=best["cores"], kind="port", host_cores=os.cpu_count(),
                sample="%d iterations of the same workload at number_samples=%d in %.1f s, oracle/svi_oracle.py on PyTorch-CPU, "
                       "%d thread(s) — the faster of the legs timed live at %s threads" % (
                           best["iterations"], n_cpu, best["seconds"], best["cores"], " and ".join(str(c) for c in legs)),
                iters_per_sec=best["iters_per_sec"], number_samples=n_cpu, runs=runs)


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


def measure(workload, args, steps, warmup, spinup_ms, world, rank, probe_rows=None, probe_iters=None, estimator=None):
    """Warm-up, spin-up and the timed region of ONE workload: exactly `steps` SVI iterations between two barriers.
    Returns (the parts of the JSON line that describe this workload, what the CPU baseline needs)."""
    import gc
    import torch
    import torch.distributed as dist
    from brancher_amd import engine, native, workloads as W
    builder, kwargs, n_per_gpu, optimizer, opt_kwargs, desc = WORKLOADS[workload]
    if args.samples:
        n_per_gpu = args.samples
    if args.dataset_size and "dataset_size" in kwargs:
        kwargs = dict(kwargs, dataset_size=args.dataset_size)
    n_global = n_per_gpu * world
    model = getattr(W, builder)(W.native_api(), **kwargs)
    estimator = estimator or args.estimator
    compiled = engine.compile_model(model, None, estimator)
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

    def train(k):
        return compiled.train(k, n_global, optimizer, seed=0, allow_persistent=allow_persistent, **opt_kwargs)

    # ---- warm-up (untimed): W steps through exactly the path that is timed (this is also where hiprtc compiles the
    #      program-specialised kernel), then — still untimed — the same path for spinup_ms so that a short timed region
    #      (the driver runs K = 20) is not measured on a GPU whose clocks have not ramped yet
    fallback = None
    failed = None
    try:
        train(max(warmup, 1))
        # (test-only: both switches must be set, so that nothing in a production environment can trip the fallback by accident)
        if world > 1 and os.environ.get("BSVI_TEST_HOOKS") == "1" and os.environ.get("BSVI_BENCH_INJECT_EXCHANGE_FAILURE") == "1" \
                and engine._exchanges:
            raise native.NativeError("injected by BSVI_BENCH_INJECT_EXCHANGE_FAILURE (tests: the fallback below)")
    except native.NativeError as err:
        if world == 1:
            raise
        failed = err
    if world > 1 and engine.collective_kind() in ("auto", "exchange"):
        # Several ranks with the one-shot exchange opted in (BSVI_COLLECTIVE=auto|exchange; the default is torch.distributed):
        # the ranks VOTE on what happened — an error on one rank only must not send that rank into other collectives than its
        # peers.  Any failure: every rank switches to RCCL through torch.distributed, says so in the line, and starts again
        # (`train` takes rank 0's parameters at every call, so the replicas restart from identical values).
        ok = torch.tensor([0.0 if failed is not None else 1.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            fallback = "one-shot exchange abandoned (%s): RCCL through torch.distributed" % str(failed or "on another rank")[:120]
            os.environ["BSVI_COLLECTIVE"] = "torch"
            os.environ["BSVI_LOOP_EXCHANGE"] = "0"
            for ex in list(engine._exchanges.values()):
                if ex:
                    ex.close()
            engine._exchanges.clear()
            getattr(compiled, "_train_plans", {}).clear()
            train(max(warmup, 1))
    elif failed is not None:
        raise failed
    barrier()
    # ---- the same K steps ONCE before any spin-up: what a caller sees on a GPU whose clocks have not ramped (reported next
    #      to the hot figure as `cold_start`; the headline stays the contract's: W warm-up steps, then K timed steps)
    cold = None
    if spinup_ms > 0:
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        c0.record()
        tc = time.perf_counter()
        train(steps)
        c1.record()
        barrier()
        cold_dt = time.perf_counter() - tc
        cold = dict(ms_per_step=cold_dt * 1e3 / steps, device_ms_per_step=c0.elapsed_time(c1) / steps,
                    value=steps / cold_dt * (n_global / 300.0), unit="it/s",
                    note="the first %d-step call after the %d warm-up steps, before the untimed spin-up" % (steps, max(warmup, 1)))
    # one collection here, so that the collector's counters start from zero: a repeat of the same call frees what it allocates,
    # the counters do not climb and no collection falls into the timed region.  (Until round 5 the collector was switched off
    # for the region; with it off the library call of a 20-iteration launch took 13-14 us instead of 10-11 —
    # tools/r5/one_shot_probe.py, profiles/r5/bench_host_overhead.txt.)
    gc.collect()
    spun = 0
    if spinup_ms > 0:
        chunk = max(steps, 200) if not hasattr(program, "enc_layers") else steps
        t_spin = time.perf_counter()
        go_on = torch.ones(1, device="cuda" if dist.get_backend() == "nccl" else "cpu") if world > 1 else None
        while True:
            more = (time.perf_counter() - t_spin) * 1e3 < spinup_ms
            if world > 1:
                # rank 0's clock decides for everybody: a rank that left the loop one call earlier than its peers would leave them
                # waiting in that call's all-reduces
                go_on[0] = 1.0 if more else 0.0
                dist.broadcast(go_on, 0)
                more = float(go_on[0]) != 0.0
            if not more:
                break
            train(chunk)
            torch.cuda.synchronize()
            spun += chunk
    barrier()

    # ---- timed region: exactly K steps
    # (torch creates the HIP event at the FIRST record: one record of each ahead of the region, so that the closing record
    #  inside it is a record and not a creation — a fresh pair cost the 20-iteration region 6-14 us of host time, same probe)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    if spinup_ms > 0:
        # ... and the last of the untimed spin-up is the region's own sequence, twice: the first K-step call bracketed this way in
        # a process takes ~12 us longer on the host than the second (cold code paths of the runtime behind a record)
        for _ in range(2):
            barrier()
            ev0.record()
            train(steps)
            ev1.record()
            spun += steps
    barrier()
    ev0.record()                    # (ahead of the host clock: the events bracket the wall region from outside)
    t0 = time.perf_counter()
    losses, finite = train(steps)
    t_launched = time.perf_counter()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    mode = compiled.last_mode

    t = torch.tensor([dt, dev_ms], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dev_ms = float(t[0]), float(t[1])
    ok = bool(torch.isfinite(losses).all()) and bool(finite.all())
    if rank != 0:
        return None, None

    iters_per_sec = steps / dt
    value = iters_per_sec * (n_global / 300.0)
    bnn = hasattr(program, "tensors") and hasattr(program, "layers")

    def make_part(roofline, geom):
        part = dict(metric="ELBO iters/sec at num_samples=%d per GPU (300-sample-equivalent iterations, whole job)" % n_per_gpu,
                    value=value, unit="it/s", n_gpus=world, steps=steps, warmup=warmup,
                    ms_per_step=dt * 1e3 / steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype="f32", data="synthetic",
                    config=dict(workload=desc, number_samples_per_gpu=n_per_gpu, number_samples_global=n_global,
                                optimizer=optimizer, **{k: v for k, v in opt_kwargs.items()},
                                estimator=estimator, mode=mode, parallelism="sample-shard x%d" % world,
                                grid=geom, untimed_spinup_iterations=spun),
                    iters_per_sec=iters_per_sec, samples_per_sec=iters_per_sec * n_global,
                    device_ms_per_step=dev_ms / steps, all_finite=ok,
                    # the timed region taken apart on the host clock: the library call(s) that launch the K steps, then the wait
                    host_launch_us=(t_launched - t0) * 1e6, host_wait_us=(dt - (t_launched - t0)) * 1e6,
                    final_loss=float(losses[-1].item()), roofline=roofline)
        if cold is not None:
            part["cold_start"] = cold
        if world > 1:
            # which all-reduce every rank took (the first SCALE run says it): torch.distributed's (RCCL under the nccl backend) unless
            # the one-shot exchange was opted in AND chosen by the self-test + vote
            used = bool(engine._exchanges.get(torch.cuda.current_device()))
            part["config"]["collective"] = ("one-shot exchange (bsvi_exchange_*)" if used else
                                            "torch.distributed all_reduce, backend %s" % dist.get_backend())
        if fallback:
            part["config"]["collective_fallback"] = fallback
        return part

    if bnn:
        # the two products of every (sample, layer-1 weight matrix): [H, P] x [P, B] forward and its weight gradient, bf16 x3 when the
        # minibatch is exactly bf16; the [C, H] layer is 2 % of that and is priced at the same peak
        L = program.layers
        flops = sum(2.0 * 2.0 * n_per_gpu * l["rows"] * l["cols"] * program.batch_size for l in L)
        tf = flops / (dev_ms * 1e-3 / steps) / 1e12
        exact = compiled.data_path() == "bf16x3"
        peak = MFMA_EXACT_PEAK_TFLOPS if exact else MFMA_F32_PEAK_TFLOPS
        noise_bytes = n_per_gpu * sum(l["rows"] * l["cols"] + (l["rows"] if l.get("bias") is not None else 0) for l in L) * 4
        roofline = dict(bound="mfma", achieved=tf, peak=peak, unit="TFLOP/s", frac=tf / peak, traffic=None,
                        kernel="bnn_draw + bnn_lower (product 1) + bnn_upper + bnn_mid (product 2) + bnn_row_sums + bnn_epilogue",
                        algorithmic_flops_per_iteration=flops, launch_ms=dev_ms / steps, data_path="bf16x3" if exact else "f32",
                        algorithmic_bytes_per_iteration=noise_bytes + program.batch_size * L[0]["cols"] * 4 + 2 * program.n_params * 4,
                        note="every sample has its OWN weight matrices (N x H x P normals per iteration: %.1f MB, drawn in registers, the "
                             "sampled weights pass through HBM once as bf16 pieces); achieved = flops of the forward products and their "
                             "weight gradients / duration of the whole iteration (6 launches); no traffic pass for this entry" % (noise_bytes / 1e6))
        part = make_part(roofline, dict(kind="bnn", layers=[(l["rows"], l["cols"]) for l in L]))
        del compiled, model
        return part, dict(builder=builder, kwargs=kwargs, n=n_per_gpu, optimizer=optimizer, opt_kwargs=opt_kwargs, dense=False, amort=False, bnn=True)
    dense = hasattr(program, "n_classes")
    amort = hasattr(program, "enc_layers")
    geom = dict(kind="dense") if dense else dict(kind="amortized") if amort else compiled.native.geometry(n_per_gpu)
    spec = None
    if not dense and not amort:
        spec = compiled.native.engine(n_per_gpu, 2 if mode == "persistent" else 1 if mode == "stepwise" else 0)
        spec = spec if spec["engine"] == "specialised" else None
    # roofline of the dominant kernel (the fused ELBO kernel; in persistent mode one launch covers all K iterations).
    # Launch duration from HIP events on the launch stream.
    alg_bytes_iter = algorithmic_bytes_per_iteration(program, n_per_gpu * (program.batch_size if amort else 1))
    traffic, traffic_how = None, None
    iters_probed = (probe_iters or {}).get(workload)
    if spec is not None:
        # the program-specialised kernel: ONE launch per bsvi_* call — the whole loop in persistent mode
        units_per_launch = steps if mode == "persistent" else 1
        launch_ms = dev_ms / (1 if mode == "persistent" else steps)
        kernel = "bsvi_spec_kernel (straight-line HIP generated from the model program, hiprtc)"
        geom = dict(engine="specialised", n_blocks=spec["n_blocks"], n_threads=spec["n_threads"],
                    lds_bytes=spec["lds_bytes"], storage="registers")
        if probe_rows:
            grid = spec["n_blocks"] * spec["n_threads"]
            if mode == "persistent":
                traffic = traffic_of(probe_rows, ["bsvi_spec_kernel"], grid=grid, last_only=True)
                traffic_how = "the timed %d-iteration launch" % steps
            else:
                traffic = traffic_of(probe_rows, ["bsvi_spec_kernel"], grid=grid, per=iters_probed or 1)
                traffic_how = "mean over the probe's %d one-iteration launches" % (iters_probed or 0)
    elif mode == "persistent":
        launch_ms, units_per_launch = dev_ms, steps
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
        if probe_rows:
            traffic = traffic_of(probe_rows, ["persistent_"], last_only=True)
            traffic_how = "the timed %d-iteration launch" % steps
    else:
        launch_ms, units_per_launch = dev_ms / steps, 1
        kernel = "bsvi::elbo_kernel<%s>" % geom.get("storage", "dense")
        V = getattr(getattr(compiled, "native", None), "_elbo_shares_set", 0)
        if not dense and not amort and V >= 2 and os.environ.get("BSVI_ELBO_SHARES", "1") != "0":
            waves = (n_per_gpu + 63) // 64
            if geom.get("storage") == "lds+lane_acc" and geom.get("lanes_per_wave") == 64 and waves * V <= 256:
                geom = dict(geom, n_blocks=waves, n_waves=1)        # one wave per workgroup (share_geometry)
            geom = dict(geom, n_blocks=geom["n_blocks"] * V, program_shares=V,
                        note="workgroup b runs share b %% %d of the model's log-prob records on sample group b / %d; "
                             "reduce_kernel adds the rows of partial sums" % (V, V))
        if probe_rows and not dense and not amort:
            traffic = traffic_of(probe_rows, ["elbo_kernel"], per=iters_probed or 1)
            traffic_how = "elbo_kernel launches of the probe / its %d iterations" % (iters_probed or 0)
    achieved = alg_bytes_iter * units_per_launch / (launch_ms * 1e-3) / 1e9
    us_iter = dev_ms * 1e3 / steps
    issue = None
    if spec is not None and spec["n_blocks"] == 1 and probe_rows:
        issue = issue_roofline(probe_rows, ["bsvi_spec_kernel"], spec["n_blocks"] * spec["n_threads"], n_per_gpu,
                               steps if mode == "persistent" else 1, us_iter, last_only=(mode == "persistent"))
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, kernel=kernel, algorithmic_bytes_per_iteration=alg_bytes_iter,
                    iterations_per_launch=units_per_launch, launch_ms=launch_ms,
                    note="latency / issue-bound workload (SURVEY §8d cfg 1-3): %.2f us per iteration; HBM is not its "
                         "roof — the noise is generated in registers, a launch fetches the parameters and writes the loss "
                         "curve — what bounds it is the serial instruction stream of one wave per sample group and the "
                         "workgroup barriers of an iteration (DESIGN.md 4.7); a launch-per-step design pays ~5-10 us of "
                         "launch latency per iteration on top" % us_iter)
    if issue is not None:
        roofline["issue"] = issue
        # (flat copies: a record that keeps only scalars of `roofline` drops the nested object)
        roofline["issue_frac"] = issue["frac"]
        roofline["issue_us_per_iteration"] = issue["issue_us_per_iteration"]
        roofline["valu_per_wave_iteration"] = issue["valu_per_wave_iteration"]
        roofline["issue_waves_on_busiest_simd"] = issue["waves_on_busiest_simd"]
        # round 5 (tools/r5/pk_rate.hip, profiles/r5/cfg1_wave_rate_notes.txt): ONE wave issues an independent VALU instruction every
        # 3.14 ns (7.5 cycles at 2.4 GHz; 4.18 ns when it depends on the one before) whether or not a second wave shares its SIMD —
        # the iteration is the instruction chain of one wave, and the figure above (two waves x 4 cycles) is the same number read
        # the other way.  wave_chain_* prices the MEAN wave's instructions at the measured per-wave rate (with the draw service the
        # launch's waves differ: sample waves ~830 + the owners' ~400 of epilogue, draw waves 470-940).
        roofline["wave_ns_per_valu"] = 3.14
        roofline["wave_chain_us_per_iteration"] = issue["valu_per_wave_iteration"] * 3.14e-3
        roofline["wave_chain_frac"] = roofline["wave_chain_us_per_iteration"] / issue["measured_us_per_iteration"]
    if dense:
        # the whole iteration (6 launches) is timed; the two MFMA GEMMs are >80 % of it (profiles/)
        flops = dense_flops_per_iteration(program, n_per_gpu)
        tf = flops / (dev_ms * 1e-3 / steps) / 1e12
        if probe_rows:
            # (the probe runs every config in one process: the products of the six-launch form, `xgemm_nt_*`, are counted only
            #  between this config's first and last `dense_` dispatch — config 5 launches the same kernels for its first layer,
            #  and the driver's line of round 3 carried them in config 4's figure: 510 MB against 304 MB measured alone.
            #  The span is found from kernels only THIS config launches: the Bayesian-neural-network entries launch dense_param_kernel too.)
            own = [r["dispatch"] for r in probe_rows if any(k in r["kernel"] for k in ("dense_head", "dense_xfwd", "dense_xbwd", "dense_forward", "dense_backward"))]
            span = [r for r in probe_rows if own and min(own) <= r["dispatch"] <= max(own)]
            traffic = traffic_of(span, ["dense_", "xgemm_nt"], per=iters_probed or 1, double_fetch=True)
            traffic_how = "all dense_* (and, six-launch form, xgemm_nt) launches of this config in the probe / its %d iterations; FETCH_SIZE doubled (gfx950 rule)" % (iters_probed or 0)
        exact = getattr(compiled, "data_path", lambda: "f32")() == "bf16x3"
        peak = MFMA_EXACT_PEAK_TFLOPS if exact else MFMA_F32_PEAK_TFLOPS
        roofline = dict(bound="mfma", achieved=tf, peak=peak, unit="TFLOP/s",
                        frac=tf / peak, traffic=traffic,
                        kernel=("dense_xfwd (draw + logits product + cross-entropy) + dense_xbwd (gradient product + reduction "
                                "against the redrawn normals)" if exact else "bsvi::dense_forward<10> + bsvi::dense_backward"),
                        algorithmic_flops_per_iteration=flops, launch_ms=dev_ms / steps, data_path="bf16x3" if exact else "f32",
                        algorithmic_bytes_per_iteration=dense_bytes_per_iteration(program, n_per_gpu)["survey"],
                        bytes_this_design_must_move_per_iteration=dense_bytes_per_iteration(program, n_per_gpu)["must_move"],
                        frac_of_f32_mfma_peak=tf / MFMA_F32_PEAK_TFLOPS,
                        note=("the minibatch is exactly bf16 (pixel counts): both products run as three bf16 MFMAs on the exact "
                              "pieces hi + mid + lo of the f32 operand; peak = dense bf16 MFMA peak / 3; achieved = GEMM flops of "
                              "one iteration / duration of the whole iteration (6 launches, the two products fused with their "
                              "neighbours: DESIGN.md 4.8); frac_of_f32_mfma_peak compares with "
                              "the f32-input MFMA kernels that serve inexact data" if exact else
                              "f32-input MFMA; achieved = GEMM flops of one iteration / duration of the whole iteration"))
    if amort:
        # the whole iteration (~23 launches) is timed; the eleven MFMA GEMMs carry the flops
        flops = amort_flops_per_iteration(program, n_per_gpu)
        tf = flops / (dev_ms * 1e-3 / steps) / 1e12
        if probe_rows:
            # (as for config 4: only from this config's first `amort_head` dispatch to the first launch of another config behind it — the Bayesian neural
            #  network's products are bsvi_amort_impl::xgemm_nt_glds_kernel launches too)
            own = [r["dispatch"] for r in probe_rows if "amort_head" in r["kernel"] or "amort_latent" in r["kernel"]]
            later = [r["dispatch"] for r in probe_rows if own and r["dispatch"] > min(own) and ("bnn_" in r["kernel"] or "dense_head" in r["kernel"] or "spec_kernel" in r["kernel"])]
            span = [r for r in probe_rows if own and min(own) <= r["dispatch"] < (min(later) if later else float("inf"))]
            traffic = traffic_of(span, ["bsvi_amort_impl"], per=iters_probed or 1, double_fetch=True)
            traffic_how = "all bsvi_amort_impl launches of this config in the probe / its %d iterations; FETCH_SIZE doubled (gfx950 rule)" % (iters_probed or 0)
        exact = getattr(compiled, "data_path", lambda: "f32")() == "bf16x3"
        rows = n_per_gpu * program.batch_size
        # flops of the forward products that read the data rows (bf16 x3 when the data is exactly bf16), the rest on the f32-input MFMA
        x_flops = sum(2.0 * rows * l.n_in * l.n_out for l in program.enc_layers if l.in_value == 0 and l.n_out > 8) if exact else 0.0
        xdw = exact and os.environ.get("BSVI_AMORT_XDW", "1") != "0"      # ... and their weight gradients (the data is an operand again)
        if xdw:
            x_flops *= 2.0
        # ... and the forward products of the other wide layers: six bf16 MFMAs on the exact pieces of BOTH operands (x6gemm_kernel)
        x6 = os.environ.get("BSVI_AMORT_X6", "1") != "0" and rows >= 256
        wide = lambda l: l.n_in >= 64 and l.n_out >= 64 and l.n_in % 4 == 0 and l.n_out % 4 == 0
        x6_layers = [l for l in program.enc_layers if l.in_value != 0 and wide(l)] + [l for l in program.dec_layers if wide(l)]
        modes = os.environ.get("BSVI_X6_MODES")          # input gradients too: unset or 3 = all of them (round 6), 1 = none
        x6_back = lambda l: modes in (None, "3")
        x6_tn = os.environ.get("BSVI_X6_TN", "1") != "0"      # round 5: their weight gradients too (x6tn_kernel)
        x6_flops = (sum(2.0 * rows * l.n_in * l.n_out for l in x6_layers) +
                    sum(2.0 * rows * l.n_in * l.n_out for l in x6_layers if x6_back(l)) +
                    (sum(2.0 * rows * l.n_in * l.n_out for l in x6_layers) if x6_tn else 0.0)) if x6 else 0.0
        roof_s = (flops - x_flops - x6_flops) / (MFMA_F32_PEAK_TFLOPS * 1e12) + x_flops / (MFMA_EXACT_PEAK_TFLOPS * 1e12) \
            + x6_flops / (MFMA_EXACT_PEAK_TFLOPS / 2.0 * 1e12)
        peak = flops / roof_s / 1e12
        roofline = dict(bound="mfma", achieved=tf, peak=peak, unit="TFLOP/s",
                        frac=tf / peak, traffic=traffic,
                        kernel="bsvi_amort_impl::gemm_kernel<1|2>" + (" + x6gemm_kernel (forward products and input gradients of the wide layers) + x6tn_kernel (their weight gradients)" if x6_flops else "<0>") + (
                            " + xgemm_nt_glds_kernel<128> (first encoder layer: forward%s)" % (" and weight gradient" if xdw else "") if exact else ""),
                        algorithmic_flops_per_iteration=flops, launch_ms=dev_ms / steps,
                        algorithmic_bytes_per_iteration=amort_bytes_per_iteration(program, n_per_gpu),
                        rows_per_iteration=rows, data_path="bf16x3" if exact else "f32",
                        frac_of_f32_mfma_peak=tf / MFMA_F32_PEAK_TFLOPS,
                        x6_flops_per_iteration=x6_flops,
                        note="f32-input MFMA (v_mfma_f32_32x32x2_f32) for the products that are not named next" + (
                             "; the forward products of the wide layers whose input is a network value, their weight gradients and their input gradients "
                             "run as SIX bf16 MFMAs on the exact pieces of both f32 operands (peak 2500 / 6 for their flops)" if x6_flops else "") + (
                             "; the forward product%s of the layer that reads the (exactly bf16) data rows run%s as three bf16 MFMAs on "
                             "the exact pieces of the f32 operand; peak = flops / (f32-input flops / 157.3 + those flops / (2500 / 3) + six-piece flops / (2500 / 6))"
                             % ((" and the weight gradient", "") if xdw else ("", "s")) if exact else "") +
                             "; achieved = GEMM flops of one iteration (forward + weight gradient + input gradient of every Linear "
                             "layer) / duration of the whole iteration")
    if traffic is not None:
        roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py before its timed " \
                                     "regions (child processes, same workloads and step counts): " + traffic_how
    part = make_part(roofline, geom)
    del compiled, model
    return part, dict(builder=builder, kwargs=kwargs, n=n_per_gpu, optimizer=optimizer, opt_kwargs=opt_kwargs,
                      dense=dense, amort=amort)


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
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="after the headline workload, time BASELINE configs 2-5 in the same process and attach them under "
                         "the `other_configs` key of the same JSON line (auto: when the headline is cfg1 on one GPU)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "on", "off"],
                    help="measure roofline.traffic live: rocprofv3 --pmc passes of this script as child processes before "
                         "the timed regions (auto: on one GPU)")
    ap.add_argument("--traffic-probe", default="", help=argparse.SUPPRESS)       # child mode: "cfg1:K:W,cfg2:K:W,..."
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

    probing = bool(args.traffic_probe)
    others = [] if probing else list(OTHER_CONFIGS) if (args.other_configs == "on" or (
        args.other_configs == "auto" and world == 1 and args.workload == "cfg1" and not args.samples)) else []
    plan = [(args.workload, args.steps, args.warmup, args.spinup_ms)] + \
           [(w, OTHER_STEPS, OTHER_WARMUP, OTHER_SPINUP_MS) for w in others]
    blackbox_plan = [(w, OTHER_STEPS, OTHER_WARMUP, OTHER_SPINUP_MS) for w in OTHER_BLACKBOX] if (others and args.estimator == "pathwise") else []
    if probing:
        plan = [(w, int(k), int(wu), 0.0) for w, k, wu in (item.split(":") for item in args.traffic_probe.split(","))]

    # ---- live HBM traffic: PMC passes of this same script as children, BEFORE this process initialises the GPU
    probe_rows, probe_note, probe_iters = None, None, {}
    if not probing and world == 1 and args.traffic != "off":
        t_probe = time.perf_counter()
        steps_of = {w: k for w, k, _, _ in plan}
        warm_of = {w: max(wu, 1) for w, _, wu, _ in plan}
        probe_rows, probe_note = run_traffic_probe([w for w, _, _, _ in plan], steps_of, warm_of, args.estimator, args.mode,
                                                   args.samples)
        probe_iters = {w: steps_of[w] + warm_of[w] for w in steps_of}
        probe_seconds = time.perf_counter() - t_probe
    elif not probing:
        probe_note = "not measured (%s)" % ("--traffic off" if args.traffic == "off" else "multi-GPU run")

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

    from brancher_amd import config
    config.set_device("cuda:%d" % device_index)

    line, baseline_of = None, None
    other_lines = {}
    for i, (workload, steps, warmup, spinup_ms) in enumerate(plan):
        if i == 0 or probing:
            part, info = measure(workload, args, steps, warmup, spinup_ms, world, rank, probe_rows, probe_iters)
            if i == 0:
                line, baseline_of = part, info
            continue
        try:        # the other configs never cost the headline its line
            part, _ = measure(workload, args, steps, warmup, spinup_ms, world, rank, probe_rows, probe_iters)
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "device_ms_per_step", "iters_per_sec", "samples_per_sec",
                    "all_finite", "final_loss", "config", "roofline", "cold_start")
            other_lines[workload] = {k: part[k] for k in keep if k in part}
        except Exception as err:      # noqa: BLE001
            other_lines[workload] = dict(error="%s: %s" % (type(err).__name__, err))
        torch.cuda.empty_cache()
    for workload, steps, warmup, spinup_ms in ([] if probing else blackbox_plan):
        try:
            part, _ = measure(workload, args, steps, warmup, spinup_ms, world, rank, None, None, estimator="blackbox")
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "device_ms_per_step", "iters_per_sec", "samples_per_sec",
                    "all_finite", "final_loss", "config", "roofline", "cold_start")
            other_lines[workload + "_blackbox"] = {k: part[k] for k in keep if k in part}
        except Exception as err:      # noqa: BLE001
            other_lines[workload + "_blackbox"] = dict(error="%s: %s" % (type(err).__name__, err))
        torch.cuda.empty_cache()

    # ---- several ranks: the SAME workload once more with the library's one-shot exchange opted in (BSVI_COLLECTIVE=auto: the in-kernel
    #      loop with the exchange inside where it serves, the exchange kernel between launches otherwise) — `value_alt` / `config.collective_alt`
    #      beside the default's RCCL figure, so that the first run on a multi-GPU node says what flipping the default would buy.  It must cost
    #      nothing but that key: the exchange's waits are bounded, `measure` votes and falls back on any rank's failure, and a WATCHDOG in every
    #      rank ends the phase after BSVI_BENCH_ALT_TIMEOUT_S (default 120): rank 0 prints the line it already has, every rank leaves with 0.
    alt = None
    if world > 1 and not probing and os.environ.get("BSVI_BENCH_ALT", "1") != "0" and engine_collective_is_default():
        import threading
        done = threading.Event()
        limit = float(os.environ.get("BSVI_BENCH_ALT_TIMEOUT_S", "120"))

        def watchdog():
            if done.wait(limit):
                return
            if rank == 0:
                line.setdefault("config", {})["collective_alt_error"] = "the BSVI_COLLECTIVE=auto phase did not finish within %.0f s: abandoned" % limit
                finish_line(line, other_lines, baseline_of, args, world, probe_rows, probe_note, None, skip_cpu=True)
                sys.stdout.flush()
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        prev = {k: os.environ.get(k) for k in ("BSVI_COLLECTIVE", "BSVI_LOOP_EXCHANGE")}
        os.environ["BSVI_COLLECTIVE"] = "auto"
        try:
            part, _ = measure(args.workload, args, args.steps, args.warmup, args.spinup_ms, world, rank)
            if rank == 0:
                alt = dict(value_alt=part["value"], ms_per_step_alt=part["ms_per_step"], device_ms_per_step_alt=part["device_ms_per_step"],
                           collective_alt=part["config"].get("collective"), mode_alt=part["config"].get("mode"),
                           collective_alt_fallback=part["config"].get("collective_fallback"))
        except Exception as err:      # noqa: BLE001  (this phase never costs the line its headline)
            alt = dict(collective_alt_error="%s: %s" % (type(err).__name__, str(err)[:200]))
        finally:
            done.set()
            for k, v in prev.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            from brancher_amd import engine as _engine
            for ex in list(_engine._exchanges.values()):
                if ex:
                    ex.close()
            _engine._exchanges.clear()
            _engine._exchange_changed()

    if rank == 0 and not probing:
        finish_line(line, other_lines, baseline_of, args, world, probe_rows, probe_note, alt,
                    probe_seconds=locals().get("probe_seconds"))
    if world > 1:
        dist.destroy_process_group()


def engine_collective_is_default():
    from brancher_amd import engine
    return engine.collective_kind() == "torch"


def finish_line(line, other_lines, baseline_of, args, world, probe_rows, probe_note, alt, skip_cpu=False, probe_seconds=None):
    """rank 0: the ONE JSON line of the run"""
    if True:
        if alt:
            cfg_keys = ("collective_alt", "mode_alt", "collective_alt_fallback", "collective_alt_error")
            for k, v in alt.items():
                if v is None:
                    continue
                if k in cfg_keys:
                    line["config"][k] = v
                else:
                    line[k] = v
        if line["roofline"].get("traffic") is None:
            line["roofline"]["traffic_note"] = probe_note or "the probe saw no dispatch of this kernel"
        elif probe_rows is not None and probe_seconds is not None:
            line["roofline"]["traffic_probe_seconds"] = probe_seconds
        if other_lines:
            line["other_configs"] = other_lines
        if world == 1 and not args.no_cpu_baseline and not skip_cpu:
            b = baseline_of
            line["cpu_baseline"] = cpu_baseline_vae(b["kwargs"], b["optimizer"], b["opt_kwargs"]) if b["amort"] else \
                cpu_baseline(b["builder"], b["kwargs"], b["n"], b["optimizer"], b["opt_kwargs"], dense=b["dense"])
            recorded = recorded_reference_timings(args.workload, b["n"], b["optimizer"])
            if recorded is not None:
                line["cpu_baseline"]["reference"] = recorded
        print(json.dumps(line))


if __name__ == "__main__":
    main()
