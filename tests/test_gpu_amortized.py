"""GPU parity of the amortised path (BASELINE config 5; csrc/amort_kernel.hip through bsvi_amort_* / bsvi_debug_gemm)
against the reference-generated fixtures tests/golden/vae_*.npz and against oracle/vae_oracle.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_err, yardstick_grad_check

pytestmark = pytest.mark.gpu
TOL = 1e-5   # north_star: 1e-5 relative on ELBO and grads


def _module_view(compiled, named):
    """{"enc/l1.weight": array} from the engine's flat buffers"""
    enc_link, dec_link = compiled.program.links
    out = {}
    seen = set()
    for tag, link in (("enc", enc_link), ("dec", dec_link)):
        for pname, par in link.named.items():
            out["%s/%s" % (tag, pname)] = named[par.name]
            seen.add(par.name)
    for par, off, size, group in compiled.program.parameters:      # a learnable prior's roots
        if par.name not in seen:
            out["prior/" + par.name] = named[par.name]
    return out


# ---- the GEMM behind every Linear layer ------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 16), (300, 70, 50), (64, 2, 512), (257, 784, 2), (1000, 136, 40), (130, 3, 3)])
def test_gemm_matches_torch(mode, shape):
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + mode)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if mode == 0:      # C = act(A B^T + bias) + post_add, rows of A gathered
        src = rnd(M + 5, K)
        rows = torch.randint(0, M + 5, (M,), generator=g).to(torch.int32).to(dev)
        Bm, bias = rnd(N, K), rnd(N)
        Cm = torch.full((M, N + 3), 7.0, device=dev)
        native.check(lib.bsvi_debug_gemm(0, ptr(src), ptr(Bm), ptr(Cm), ptr(rows), M, N, K, K, K, N + 3, ptr(bias), 0, 2, 0.1, 0, None))
        ref = torch.nn.functional.softplus(src[rows.long()].double() @ Bm.double().T + bias.double()) + 0.1
        got = Cm[:, :N]
        assert torch.all(Cm[:, N:] == 7.0)
    elif mode == 1:    # C = (A B) * relu'(Y), accumulated on top of C
        A, Bm, Y = rnd(M, K), rnd(K, N), rnd(M, N)
        C0 = rnd(M, N)
        Cm = C0.clone()
        native.check(lib.bsvi_debug_gemm(1, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, K, N, N, ptr(Y), N, 1, 0.0, 1, None))
        ref = C0.double() + (A.double() @ Bm.double()) * (Y > 0).double()
        got = Cm
    else:              # C += A^T B with K = rows split over workgroups, rows of B gathered
        A = rnd(K, M)
        src = rnd(K + 3, N)
        rows = torch.randint(0, K + 3, (K,), generator=g).to(torch.int32).to(dev)
        Cm = torch.zeros(M, N, device=dev)
        colsum = torch.zeros(M, device=dev)
        native.check(lib.bsvi_debug_gemm(2, ptr(A), ptr(src), ptr(Cm), ptr(rows), M, N, K, M, N, N, ptr(colsum), 0, 0, 0.0, 0, None))
        ref = A.double().T @ src[rows.long()].double()
        got = Cm
        assert ((colsum.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max()).item() < 2e-6
    torch.cuda.synchronize()
    err = (got.double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    assert err < 2e-6, err


@pytest.mark.parametrize("mode", [5, 6])
@pytest.mark.parametrize("shape", [(256, 128, 64), (300, 200, 100), (1000, 784, 512), (25600, 512, 256), (25600, 256, 784),
                                   (25600, 784, 512), (4100, 72, 260)])
def test_six_piece_products_on_the_bf16_matrix_cores_match_f32_accuracy(mode, shape):
    """x6gemm_kernel (the wide layers of the amortised path from 256 rows): an f32 x f32 product as six bf16 MFMAs on the exact
    pieces hi + mid + lo of both operands.  Against the product in double precision, with torch's own f32 matmul as the
    yardstick: at most twice its error (or 2e-6 of the largest output) — shapes with tails in every dimension (K = 100, 260,
    784 are not multiples of the k step 32; N = 200, 72, 784 not of the tile; M = 300, 4100 not either)."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K + mode)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if mode == 5:
        A, Bm, bias = rnd(M, K), rnd(N, K), rnd(N)
        Cm = torch.full((M, N + 4), 7.0, device=dev)
        native.check(lib.bsvi_debug_gemm(5, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, K, K, N + 4, ptr(bias), 0, 0, 0.0, 0, None))
        ref = A.double() @ Bm.double().T + bias.double()
        f32 = A @ Bm.T + bias
        got = Cm[:, :N]
        assert torch.all(Cm[:, N:] == 7.0)
    else:
        A, Bm, Y, C0 = rnd(M, K), rnd(K, N), rnd(M, N), rnd(M, N)
        Cm = C0.clone()
        native.check(lib.bsvi_debug_gemm(6, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, K, N, N, ptr(Y), N, 1, 0.0, 1, None))
        ref = C0.double() + (A.double() @ Bm.double()) * (Y > 0).double()
        f32 = C0 + (A @ Bm) * (Y > 0).float()
        got = Cm
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item() / scale
    yard = (f32.double() - ref).abs().max().item() / scale
    assert err <= max(2.0 * yard, 2e-6), (err, yard)


@pytest.mark.parametrize("shape", [(784, 256, 25600), (512, 256, 25600), (256, 512, 25600), (200, 72, 17000), (300, 130, 4100), (128, 128, 33),
                                   (64, 260, 1000)])
def test_six_piece_weight_gradient_matches_f32_accuracy(shape):
    """x6tn_kernel (round 5: the weight gradients of the amortised path's wide layers): C[m][n] = sum_k A[k][m] B[k][n] with BOTH
    operands f32 activations — split into exact bf16 pieces and transposed on the way into LDS, six MFMAs per k chunk, one partial per
    slice of the rows, the slices added in order; the column sums of A (the bias gradient) ride along.  Same bound as the forward
    form: at most twice the error of torch's own f32 matmul against the product in double precision (or 2e-6 of the largest output);
    cfg 5's three layer shapes at 25 600 rows, and shapes with tails in every dimension; bit-identical call to call."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    ldm = M + 4                                   # (dY lives in a wider value buffer)
    Abuf, Bm = rnd(K, ldm), rnd(K, N)
    A = Abuf[:, :M]
    outs = []
    for _ in range(2):
        Cm, colsum = torch.full((M, N + 4), 7.0, device=dev), torch.zeros(M, device=dev)
        native.check(lib.bsvi_debug_gemm(7, ptr(Abuf), ptr(Bm), ptr(Cm), None, M, N, K, ldm, N, N + 4, ptr(colsum), 0, 0, 0.0, 0, None))
        torch.cuda.synchronize()
        outs.append((Cm, colsum))
    (Cm, colsum), (Cm2, colsum2) = outs
    assert torch.equal(Cm, Cm2) and torch.equal(colsum, colsum2)
    assert torch.all(Cm[:, N:] == 7.0)
    ref = A.double().T @ Bm.double()
    f32 = A.T @ Bm
    scale = ref.abs().max().item()
    err = (Cm[:, :N].double() - ref).abs().max().item() / scale
    yard = (f32.double() - ref).abs().max().item() / scale
    assert err <= max(2.0 * yard, 2e-6), (err, yard)
    cs = A.double().sum(0)
    assert ((colsum.double() - cs).abs().max() / cs.abs().max()).item() < 2e-6
    # the data layer's form: B's rows gathered from a dataset of exactly-bf16 values (pixel counts) — one piece, three products
    n_src = 3000
    src = torch.randint(0, 256, (n_src, N), generator=g).float().to(dev)
    rows = torch.randint(0, n_src, (K,), generator=g, dtype=torch.int32).to(dev)
    Cx, colx = torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)
    native.check(lib.bsvi_debug_gemm(7, ptr(Abuf), ptr(src), ptr(Cx), ptr(rows), M, N, K, ldm, N, N, ptr(colx), 0, 0, 0.0, 0, None))
    torch.cuda.synchronize()
    Bg = src[rows.long()]
    refx = A.double().T @ Bg.double()
    sx = refx.abs().max().item()
    errx = (Cx.double() - refx).abs().max().item() / sx
    yardx = ((A.T @ Bg).double() - refx).abs().max().item() / sx
    assert errx <= max(2.0 * yardx, 2e-6), (errx, yardx)
    assert torch.equal(colx, colsum)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(25600, 256, 784), (25600, 512, 256), (17000, 200, 72), (25600, 784, 256), (25000, 256, 512),
                                   (9000, 300, 48)])
def test_gemm_tall_products_at_config5_rows(mode, shape):
    """tall products at the row counts of BASELINE config 5 (edge tiles along M and N): against torch in double precision, and
    bit-identical call after call."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + N + K + mode)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    pad = lambda n: (n + 3) // 4 * 4
    if mode == 0:
        A, Bm, bias = rnd(M, pad(K)), rnd(N, pad(K)), rnd(N)
        run = lambda Cm: native.check(lib.bsvi_debug_gemm(0, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, pad(K), pad(K), N, ptr(bias), 0, 1, 0.0, 0, None))
        ref = torch.relu(A[:, :K].double() @ Bm[:, :K].double().T + bias.double())
    else:
        A, Bm, Y = rnd(M, pad(K)), rnd(K, N), rnd(M, N)
        run = lambda Cm: native.check(lib.bsvi_debug_gemm(1, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, pad(K), N, N, ptr(Y), N, 1, 0.0, 0, None))
        ref = (A[:, :K].double() @ Bm.double()) * (Y > 0).double()
    outs = []
    for _ in range(3):
        Cm = torch.full((M, N), 3.0, device=dev)
        run(Cm)
        torch.cuda.synchronize()
        outs.append(Cm)
    err = (outs[0].double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    assert err < 2e-6, err
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_gemm_tall_split_k():
    """backward-weight shape of the workload: K = all rows (tens of thousands), small output"""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    K, M, N = 40000, 72, 200
    g = torch.Generator(device="cpu").manual_seed(5)
    A, Bm = torch.randn(K, M, generator=g).to(dev), torch.randn(K, N, generator=g).to(dev)
    Cm = torch.zeros(M, N, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    native.check(lib.bsvi_debug_gemm(2, ptr(A), ptr(Bm), ptr(Cm), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, None))
    ref = A.double().T @ Bm.double()
    assert ((Cm.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("kind", ["binary", "counts"])
@pytest.mark.parametrize("shape", [(128, 128, 32), (300, 70, 50), (1000, 136, 784), (25600, 256, 784), (513, 257, 150)])
def test_exact_data_forward_product_on_the_bf16_matrix_cores(kind, shape):
    """The layer that reads the data rows when every data value is exactly a bf16 (binarised images, pixel counts 0..255):
    x W^T as three bf16 MFMAs on the exact pieces of W (xgemm_nt_kernel).  Against torch in double precision at the
    tolerance of the f32-input kernel, against that kernel itself, and bit-identical call after call."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 5 * K)
    n_src = M // 2 + 7
    if kind == "binary":
        src = (torch.rand(n_src, K, generator=g) > 0.5).float().to(dev)
    else:
        src = torch.randint(0, 256, (n_src, K), generator=g).float().to(dev)
    rows = torch.randint(0, n_src, (M,), generator=g).to(torch.int32).to(dev)
    Bm, bias = (torch.randn(N, K, generator=g) * (1.0 if kind == "binary" else 1.0 / 64)).to(dev), torch.randn(N, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    ref = torch.relu(src[rows.long()].double() @ Bm.double().T + bias.double())
    outs = []
    for _ in range(2):
        Cm = torch.full((M, N + 3), 7.0, device=dev)
        native.check(lib.bsvi_debug_gemm(3, ptr(src), ptr(Bm), ptr(Cm), ptr(rows), M, N, K, K, K, N + 3, ptr(bias), 0, 1, 0.0, n_src, None))
        torch.cuda.synchronize()
        assert torch.all(Cm[:, N:] == 7.0)
        outs.append(Cm[:, :N].clone())
    scale = ref.abs().max().item() + 1e-12
    assert (outs[0].double() - ref).abs().max().item() / scale < 2e-6
    assert torch.equal(outs[0], outs[1])
    Cf = torch.zeros(M, N, device=dev)
    native.check(lib.bsvi_debug_gemm(0, ptr(src), ptr(Bm), ptr(Cf), ptr(rows), M, N, K, K, K, N, ptr(bias), 0, 1, 0.0, 0, None))
    torch.cuda.synchronize()
    assert (outs[0].double() - Cf.double()).abs().max().item() / scale < 2e-6


@pytest.mark.parametrize("kind", ["binary", "counts"])
@pytest.mark.parametrize("shape", [(64, 64, 64), (70, 150, 300), (256, 784, 25600), (136, 100, 1000), (257, 96, 4133)])
def test_exact_data_weight_gradient_on_the_bf16_matrix_cores(kind, shape):
    """dW = dY^T x[rows] of the layer that reads the data rows, x exact in bf16: the three bf16 pieces of dY, transposed,
    times the transposed minibatch rows, split over the rows (dy_split_t_kernel, xt_gather_kernel, the split-k form of
    xgemm_nt_glds_kernel).  Against torch in double precision at the tolerance of the f32-input kernel, against that
    kernel, the column sums of dY (the bias gradient), and bit-identical call after call."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    M, N, K = shape                      # dW is [M = n_out][N = n_in], K rows
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 5 * K)
    n_src = K // 2 + 7
    if kind == "binary":
        src = (torch.rand(n_src, N, generator=g) > 0.5).float().to(dev)
    else:
        src = torch.randint(0, 256, (n_src, N), generator=g).float().to(dev)
    rows = torch.randint(0, n_src, (K,), generator=g).to(torch.int32).to(dev)
    ld = M + 4
    dY = torch.randn(K, ld, generator=g).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    ref = dY[:, :M].double().T @ src[rows.long()].double()
    ref_b = dY[:, :M].double().sum(0)
    outs = []
    for _ in range(2):
        Cm = torch.full((M, N + 4), 7.0, device=dev)
        colsum = torch.zeros(M, device=dev)
        native.check(lib.bsvi_debug_gemm(4, ptr(dY), ptr(src), ptr(Cm), ptr(rows), M, N, K, ld, N, N + 4, ptr(colsum), 0, 0, 0.0, n_src, None))
        torch.cuda.synchronize()
        assert torch.all(Cm[:, N:] == 7.0)
        outs.append((Cm[:, :N].clone(), colsum))
    scale = ref.abs().max().item() + 1e-12
    assert (outs[0][0].double() - ref).abs().max().item() / scale < 2e-6
    assert (outs[0][1].double() - ref_b).abs().max().item() / ref_b.abs().max().item() < 2e-6
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    Cf = torch.zeros(M, N, device=dev)
    native.check(lib.bsvi_debug_gemm(2, ptr(dY), ptr(src), ptr(Cf), ptr(rows), M, N, K, ld, N, N, None, 0, 0, 0.0, 0, None))
    torch.cuda.synchronize()
    assert (outs[0][0].double() - Cf.double()).abs().max().item() / scale < 2e-6


def test_inexact_data_stays_on_the_f32_kernel():
    """a dataset with values that are not bf16 numbers must not take the bf16 path: the compiled object says which one it runs"""
    from brancher_amd import engine, workloads as W
    import brancher_amd.workloads as Wm
    api = W.native_api()
    kw = dict(dataset_size=80, batch_size=10, n_features=96, hidden1=160, hidden2=48, seed=5)
    exact = engine.compile_model(W.build_vae(api, **kw), None, "pathwise")
    assert exact.data_path() == "bf16x3"
    original = Wm.vae_data
    try:
        Wm.vae_data = lambda ds, nf, seed=0, real=False: original(ds, nf, seed).astype("float32") * 0.3       # 0.3 is not a bf16 number
        inexact = engine.compile_model(W.build_vae(api, **kw), None, "pathwise")
    finally:
        Wm.vae_data = original
    assert inexact.data_path() == "f32"


# ---- the ELBO gradient against the reference's own outputs ------------------------------------------------------
@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_vae_golden_loss_and_grads(vae_golden, estimator):
    from brancher_amd import engine
    g = vae_golden
    model = g.build()
    compiled = engine.compile_model(model, model.posterior_model, estimator)
    assert type(compiled).__name__ == "CompiledAmortized"
    res = compiled.evaluate(g.N, noise=g.data["noise/z"], minibatch=g.data["minibatch/x"], want_fvalues=True)
    ref_loss = float(g.data["loss_" + estimator])
    assert abs(float(res["loss"]) - ref_loss) <= TOL * abs(ref_loss)
    assert float(res["finite"]) == 1.0
    grads = _module_view(compiled, compiled.named_grads())
    ref_grads = g.group("grad_%s/" % estimator)
    scale = max(np.abs(v).max() for v in ref_grads.values())
    for name, g_ref in ref_grads.items():
        assert np.abs(grads[name] - g_ref).max() <= TOL * scale, name
    f_ref = (g.data["lp"] + g.data["H"]).reshape(-1)
    assert rel_err(res["f"].cpu().numpy(), f_ref) <= TOL
    assert rel_err(res["logq"].cpu().numpy(), g.data["lq"].reshape(-1)) <= TOL


@pytest.mark.parametrize("which", ["baseline", "softmax"])
def test_vae_golden_user_defined_estimators(vae_golden, which):
    """The GradientEstimator seam (gradient_estimators.py:17-26) on the amortised path: the SAME subclass bodies the real
    reference ran (workloads.custom_estimators, oracle/gen_golden_vae.py) around two passes of the path — per-row f and
    log q, then one weight per row on the gradient seeds (bsvi_amort_args::f_weight_dev / q_weight_dev) — against the
    reference's loss and gradients on its recorded draws and minibatches."""
    from brancher_amd import engine, gradient_estimators as ge, workloads as W
    g = vae_golden
    model = g.build()
    cls = W.custom_estimators(ge)[which]
    value = engine.custom_estimator_loss(model, model.posterior_model, cls, g.N, noise=g.data["noise/z"],
                                         minibatch=g.data["minibatch/x"])
    compiled = value.compiled
    assert type(compiled).__name__ == "CompiledAmortized"
    ref_loss = float(g.data["loss_custom_" + which])
    assert abs(-float(value.detach().cpu()) - ref_loss) <= TOL * abs(ref_loss)
    grads = _module_view(compiled, compiled.named_grads())
    ref_grads = g.group("grad_custom_%s/" % which)
    # the baseline estimator multiplies grad log q by f - mean(f): rows whose f is ~500 and whose differences are ~10, so the
    # reference's own fp32 gradients sit a few 1e-5 from the fp64 ones.  Yardstick: as close to the fp64 oracle as the fp32
    # reference is (x4), or the 1e-5 of every other workload.
    from oracle.vae_oracle import VaeOracle
    fn = {"baseline": lambda f, lq: (lq * (f - f.mean()).detach() + f).mean(),
          "softmax": lambda f, lq: (torch.softmax(0.1 * f.detach().reshape(-1), dim=0).reshape(f.shape) * f).sum()}[which]
    exact = VaeOracle(g.build(), dtype=torch.float64).loss_and_grads(g.data["minibatch/x"], g.data["noise/z"], fn)["grads"]
    scale = max(np.abs(v).max() for v in exact.values())
    for name, g_ref in ref_grads.items():
        err, err_ref = np.abs(grads[name] - exact[name]).max(), np.abs(g_ref - exact[name]).max()
        assert err <= max(4 * err_ref, TOL * scale), (name, err, err_ref)


def test_vae_seam_estimators_reproduce_the_builtin_programs_at_config5_size():
    """BASELINE config 5 at full size (784-512-256-2 / 2-256-512-784, minibatch 100, 8 samples): BlackBox and Pathwise spelled
    out by a USER as GradientEstimator subclasses (the reference's own bodies, gradient_estimators.py:29-44) give the loss
    and every gradient of the built-in programs on the same in-kernel draw and minibatches; the weighted pass is linear in
    its weights; half a pair of weights is refused at the boundary."""
    import ctypes as C
    from brancher_amd import engine, gradient_estimators as ge, native, workloads as W

    class MyBlackBox(ge.GradientEstimator):
        def __call__(self, n_samples):
            samples = self.sampler._get_sample(n_samples, differentiable=False)
            samples.update(self.empirical_samples)
            variational_loss = self.sampler.calculate_log_probability(samples) * (self.function(samples).detach())
            return (variational_loss + self.function(samples)).mean()

    class MyPathwise(ge.GradientEstimator):
        def __call__(self, n_samples):
            samples = self.sampler._get_sample(n_samples, differentiable=True)
            samples.update(self.empirical_samples)
            return self.function(samples).mean()

    N, B = 8, 100
    for cls, builtin in ((MyBlackBox, "blackbox"), (MyPathwise, "pathwise")):
        model = W.build_vae(W.native_api(), dataset_size=3000, batch_size=B, n_features=784, hidden1=512, hidden2=256, seed=3)
        c = engine.compile_model(model, None, "blackbox")
        offset = c.iteration
        value = engine.custom_estimator_loss(model, model.posterior_model, cls, N)
        got = value.compiled.out[engine.OUT_HEADER:].cpu().numpy().copy()
        ref_c = engine.compile_model(model, None, builtin)
        ref = ref_c.evaluate(N, seed=None, offset=offset)
        assert abs(-float(value.detach().cpu()) - float(ref["loss"])) <= TOL * abs(float(ref["loss"]))
        want = ref_c.out[engine.OUT_HEADER:].cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    gen = torch.Generator().manual_seed(5)
    a1, a2, b1, b2 = (torch.randn(N, B, generator=gen).to(c.device) for _ in range(4))
    blocks = []
    for a, b in ((a1, b1), (a2, b2), (a1 + a2, b1 + b2)):
        c.evaluate_weighted(N, a, b, 21, 5)
        torch.cuda.synchronize()
        blocks.append(c.out[engine.OUT_HEADER:].cpu().numpy().astype(np.float64))
    scale = np.abs(blocks[2]).max()
    assert scale > 0 and np.abs(blocks[0] + blocks[1] - blocks[2]).max() <= 2e-5 * scale
    args = c._args(N, N, 0, None, None, 21, 5, f_weight=a1.reshape(-1).contiguous())
    assert c.lib.bsvi_amort_fwd_bwd(c.handle, C.byref(args)) == -1           # BSVI_ERR_INVALID


def test_public_loop_with_a_user_defined_estimator_on_the_dense_and_amortised_paths():
    """`perform_inference(..., ReverseKL(gradient_estimator=<user class>))` (inference.py:52-111) on a dense-link model and
    on a VAE: every iteration is two passes of the path around the user's torch code, the optimizer step on the device."""
    from brancher_amd import gradient_estimators as ge, inference, workloads as W
    est = W.custom_estimators(ge)["baseline"]
    api = W.native_api()
    torch.manual_seed(1234)                  # (the Philox key of the draws is torch's initial seed: not whatever an earlier test left)
    np.random.seed(1234)
    for model, n, lr in ((W.build_logistic_regression(api, dataset_size=200, batch_size=50, n_features=16, n_classes=3), 32, 0.05),
                         (W.build_vae(api, dataset_size=200, batch_size=20, n_features=40, hidden1=24, hidden2=16, seed=1), 8, 0.01)):
        inference.perform_inference(model, inference_method=inference.ReverseKL(gradient_estimator=est),
                                    number_iterations=120, number_samples=n, optimizer="Adam", lr=lr)
        curve = model.diagnostics["loss curve"]
        assert len(curve) == 120 and np.all(np.isfinite(curve)) and curve[-20:].mean() < curve[:20].mean(), curve[::10]


def test_vae_golden_trajectory(vae_golden):
    from brancher_amd import engine
    g = vae_golden
    tr = g.meta["trajectory"]
    if tr is None:
        pytest.skip("no trajectory in this fixture")
    model = g.build()
    compiled = engine.compile_model(model, model.posterior_model, "pathwise")
    losses, finite = compiled.train(tr["iters"], tr["n"], tr["optimizer"], noise_seq=list(g.data["traj/noise/z"]),
                                    minibatch_seq=list(g.data["traj/minibatch/x"]), **g.opt_kwargs())
    assert rel_err(losses.cpu().numpy(), g.data["traj/losses"]) <= TOL
    assert float(finite.min()) == 1.0
    params = _module_view(compiled, compiled.named_params())
    # Adam divides by sqrt(v): an element whose gradient is rounding noise (a ReLU unit that is almost dead on this
    # minibatch) moves by a step whose size does not depend on the gradient's size, so the difference between THIS
    # summation order and torch's shows there (run to run the engine's own result is bit-identical, tested below).
    # Bound: 1e-5 relative, plus 1 % of the largest distance Adam can move a parameter in these iterations.
    slack = 0.01 * tr["lr"] * tr["iters"] if tr["optimizer"] == "Adam" else 0.0
    for name, ref in g.group("traj/param_after/").items():
        assert np.abs(params[name] - ref).max() <= 1e-5 * (1 + np.abs(ref).max()) + slack, name
    # the trained tensors go back into the user's torch modules
    compiled.sync_modules()
    enc = compiled.program.links[0].module
    assert np.array_equal(enc.l1.weight.detach().numpy(), params["enc/l1.weight"])


# ---- against the oracle at the example's full architecture -------------------------------------------------------
@pytest.mark.parametrize("estimator", ["pathwise", "blackbox"])
def test_vae_full_architecture_matches_oracle(estimator):
    from brancher_amd import engine, workloads as W
    from oracle.vae_oracle import VaeOracle
    N, B, DS = 6, 50, 300
    model = W.build_vae(W.native_api(), dataset_size=DS, batch_size=B, n_features=784, hidden1=512, hidden2=256, seed=3)
    rng = np.random.RandomState(11)
    rows = np.stack([rng.choice(DS, B, replace=False) for _ in range(N)])
    eps = rng.randn(N, B, 2).astype(np.float32)
    ref = VaeOracle(model, dtype=torch.float64).loss_and_grads(rows, eps, estimator)
    compiled = engine.compile_model(model, model.posterior_model, estimator)
    res = compiled.evaluate(N, noise=eps, minibatch=rows)
    assert abs(float(res["loss"]) - ref["loss"]) <= TOL * abs(ref["loss"])
    grads = _module_view(compiled, compiled.named_grads())
    scale = max(np.abs(v).max() for v in ref["grads"].values())
    for name, g_ref in ref["grads"].items():
        assert np.abs(grads[name] - g_ref).max() <= TOL * scale, name


def test_vae_device_rng_properties():
    """in-kernel minibatches and noise: distinct rows per sample, different per sample and per iteration, standard
    normal eps, reproducible for a given (seed, offset); the loss agrees with the oracle fed the same draws"""
    from brancher_amd import engine, workloads as W
    from oracle.vae_oracle import VaeOracle
    N, B, DS = 64, 20, 97
    model = W.build_vae(W.native_api(), dataset_size=DS, batch_size=B, n_features=40, hidden1=24, hidden2=16, seed=1)
    compiled = engine.compile_model(model, model.posterior_model, "pathwise")
    r1 = compiled.evaluate(N, seed=5, offset=0, want_noise=True, want_indices=True)
    idx, eps, loss1 = r1["indices"].cpu().numpy(), r1["noise"].cpu().numpy(), float(r1["loss"])
    assert idx.shape == (N, B) and idx.min() >= 0 and idx.max() < DS
    assert all(len(set(row)) == B for row in idx)
    assert len({tuple(row) for row in idx}) == N
    assert abs(eps.mean()) < 0.1 and abs(eps.std() - 1) < 0.1
    r2 = compiled.evaluate(N, seed=5, offset=0, want_indices=True)
    assert float(r2["loss"]) == loss1                               # fixed-order sums: bit-reproducible
    r3 = compiled.evaluate(N, seed=5, offset=1, want_indices=True)
    assert not np.array_equal(r3["indices"].cpu().numpy(), idx)
    ref = VaeOracle(model, dtype=torch.float64).loss_and_grads(idx, eps.reshape(N, B, 2), "pathwise")
    assert abs(loss1 - ref["loss"]) <= TOL * abs(ref["loss"])


def test_vae_frame_level_samplers_match_the_reference_structure():
    """`model.get_sample` (the posterior-predictive step of examples/VAE_playground.py:90-103) and
    `model.get_posterior_sample` (variables.py:796-812) of an amortised model, against what the real reference returns
    (tests/golden/frames/vae_frames.npz, oracle/gen_golden_vae_frames.py): columns, cell shapes, and for a given latent
    value the decoder output itself."""
    import json
    import os
    from conftest import GOLDEN
    from brancher_amd import workloads as W
    fx = np.load(os.path.join(GOLDEN, "frames", "vae_frames.npz"))
    meta = json.loads(str(fx["meta"]))
    model = W.build_vae(W.native_api(), **meta["kwargs"])
    roots = {"x_total_count", "z_scale", "z_loc"}                 # (frames here carry no root columns)
    # ancestral sample of the joint model
    frame = model.get_sample(2)
    assert set(frame.columns) == set(meta["get_sample_columns"]) - roots and len(frame) == 2
    raw = meta["get_sample_raw"]
    assert frame["z"].values[0].shape == tuple(raw["z"][2:])
    assert frame["x"].values[1].shape == tuple(raw["x"][2:]) and set(np.unique(frame["x"].values[1])) <= {0.0, 1.0}
    assert frame["decoder_output"].values[0]["mean"].shape == tuple(raw["decoder_output"]["mean"][2:])
    # a given latent value: the decoder output is the reference's
    z = model.get_variable("z")
    given = model.get_sample(1, input_values={z: fx["z_value"]})
    assert set(given.columns) == set(meta["get_sample_given_columns"]) - roots
    assert np.array_equal(np.asarray(given["z"].values[0], dtype=np.float32), fx["z_given_cell"])
    got = np.asarray(given["decoder_output"].values[0]["mean"])
    assert np.abs(got - fx["decoder_mean_given_z"]).max() <= 1e-5 * (1 + np.abs(fx["decoder_mean_given_z"]).max())
    assert np.asarray(given["x"].values[0]).shape == fx["x_given_cell"].shape
    # posterior sample: a minibatch per sample and z from the encoder, keyed like the joint model's variables
    N = 200
    post = model._get_posterior_sample(N)
    by_name = {v.name: t for v, t in post.items()}
    assert set(by_name) == set(meta["get_posterior_sample_raw"])
    B, P, Dz = meta["kwargs"]["batch_size"], meta["kwargs"]["n_features"], 2
    x, zs = by_name["x"].cpu().numpy(), by_name["z"].cpu().numpy()
    assert x.shape == (N, B, P, 1) and zs.shape == (N, B, Dz)
    assert all(v.name in ("x", "z") and v in model.flatten() for v in post)
    data = W.vae_data(meta["kwargs"]["dataset_size"], P)[..., 0].astype(np.float32)
    for n in range(0, N, 37):
        which = [int(np.flatnonzero((data == x[n, b, :, 0]).all(1))[0]) for b in range(B)]
        assert len(set(which)) == B                                # distinct rows, like np.random.choice(replace=False)
    enc = model.vae_modules[0]
    with torch.no_grad():
        out = enc(torch.from_numpy(x.reshape(N * B, P, 1)))
    resid = (zs.reshape(N * B, Dz) - out["mean"].numpy()) / out["sd"].numpy()
    assert abs(resid.mean()) < 0.1 and 0.9 < resid.std() < 1.1
    pframe = model.get_posterior_sample(3)
    assert list(pframe.columns) == meta["get_posterior_sample_columns"] and len(pframe) == 3
    assert pframe["x"].values[0].shape == (B, P, 1) and pframe["z"].values[2].shape == (B, Dz)


def test_vae_perform_inference_api():
    """the user-facing loop of examples/VAE_playground.py:81-87 runs on the native engine and the loss goes down"""
    from brancher_amd import inference, workloads as W
    from brancher_amd.gradient_estimators import PathwiseDerivativeEstimator
    model = W.build_vae(W.native_api(), dataset_size=200, batch_size=25, n_features=64, hidden1=48, hidden2=32, seed=2)
    inference.perform_inference(model, inference_method=inference.ReverseKL(gradient_estimator=PathwiseDerivativeEstimator),
                                number_iterations=150, number_samples=4, optimizer="Adam", lr=0.005)
    curve = model.diagnostics["loss curve"]
    assert len(curve) == 150 and np.all(np.isfinite(curve))
    assert curve[-20:].mean() < curve[:20].mean()


def test_vae_sharded_path_matches_single():
    """the sequence a rank runs under torch.distributed (fwd_bwd -> all-reduce -> fused finalize step), on one GPU"""
    from brancher_amd import engine, workloads as W
    kw = dict(dataset_size=60, batch_size=10, n_features=30, hidden1=20, hidden2=12, seed=4)
    rng = np.random.RandomState(0)
    rows = [np.stack([rng.choice(60, 10, replace=False) for _ in range(8)]) for _ in range(4)]
    eps = [rng.randn(8, 10, 2).astype(np.float32) for _ in range(4)]
    out = []
    for forced in (False, True):
        model = W.build_vae(W.native_api(), **kw)
        compiled = engine.compile_model(model, model.posterior_model, "pathwise")
        losses, _ = compiled.train(4, 8, "Adam", noise_seq=eps, minibatch_seq=rows, lr=1e-2, _force_sharded_path=forced)
        out.append((losses.cpu().numpy(), compiled.params.cpu().numpy()))
    assert rel_err(out[1][0], out[0][0]) <= 1e-6
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-6


@pytest.mark.parametrize("kw,n", [(dict(dataset_size=60, batch_size=10, n_features=30, hidden1=20, hidden2=12, seed=4), 8),
                                  (dict(dataset_size=400, batch_size=50, n_features=784, hidden1=512, hidden2=256, seed=3), 16)])
def test_decoder_bucket_is_reduced_early_and_changes_nothing(kw, n, monkeypatch):
    """on several ranks the decoder's range of the output block is reduced on a bucket stream as soon as the
    decoder's weight gradients are in flight, and all-reduced there beside the encoder's backward pass (bsvi_amort_bucket /
    bsvi_amort_set_bucket_stream).  The rank's sequence on one GPU (`_force_sharded_path`: the all-reduce of one rank is the
    identity), with and without the bucket: the same sums from the same partials — bit-identical trajectories."""
    from brancher_amd import engine, workloads as W
    out = []
    for buckets in ("1", "0"):
        monkeypatch.setenv("BSVI_AMORT_BUCKETS", buckets)
        model = W.build_vae(W.native_api(), **kw)
        compiled = engine.compile_model(model, model.posterior_model, "pathwise")
        losses, finite = compiled.train(6, n, "Adam", seed=7, lr=1e-3, _force_sharded_path=True)
        torch.cuda.synchronize()
        assert bool(finite.all())
        assert ("+bucket" in compiled.last_mode) == (buckets == "1"), compiled.last_mode
        out.append((losses.cpu().numpy(), compiled.params.cpu().numpy().copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_vae_decode_and_encode_match_the_torch_modules():
    """posterior-predictive step of examples/VAE_playground.py:90-103: the networks applied to caller-supplied rows"""
    from brancher_amd import engine, workloads as W
    model = W.build_vae(W.native_api(), dataset_size=50, batch_size=10, n_features=150, hidden1=136, hidden2=40, latent_size=3, seed=6)
    compiled = engine.compile_model(model, model.posterior_model, "pathwise")
    enc, dec = model.vae_modules
    rng = np.random.RandomState(3)
    z = rng.randn(333, 3).astype(np.float32)
    x = (rng.rand(77, 150) > 0.5).astype(np.float32)
    with torch.no_grad():
        ref_logits = dec(torch.from_numpy(z).double() if False else torch.from_numpy(z))["mean"].numpy()
        ref_enc = enc(torch.from_numpy(x).unsqueeze(-1))
    assert rel_err(compiled.decode(z).cpu().numpy(), ref_logits) <= 2e-6
    assert rel_err(compiled.encode(x, "mean").cpu().numpy(), ref_enc["mean"].numpy()) <= 2e-6
    assert rel_err(compiled.encode(x, "sd").cpu().numpy(), ref_enc["sd"].numpy()) <= 2e-6


def test_vae_full_size_shard_linearity():
    """BASELINE config 5 at its per-GPU size (number_samples=256 x batch 100 = 25 600 rows, 784-256-512-(2,2)): the sums
    of one evaluation equal the sums of its two half shards replayed on the same minibatches and noise — the property
    the multi-GPU step relies on — and the per-row values agree row by row."""
    from brancher_amd import engine, workloads as W
    N, B = 256, 100
    model = W.build_vae(W.native_api(), dataset_size=4000, batch_size=B, n_features=784, hidden1=512, hidden2=256, seed=7)
    c = engine.compile_model(model, model.posterior_model, "blackbox")
    full = c.evaluate(N, seed=3, offset=0, want_noise=True, want_indices=True, want_fvalues=True)
    loss, grads = float(full["loss"]), full["grads"].clone()
    f_rows, eps, rows = full["f"].clone(), full["noise"].cpu().numpy().reshape(N, B, 2), full["indices"].cpu().numpy()
    assert float(full["finite"]) == 1.0 and all(len(set(r)) == B for r in rows[:16])
    half = N // 2
    sums, parts = 0.0, []
    gsum = torch.zeros_like(grads)
    for h in range(2):
        sl = slice(h * half, (h + 1) * half)
        r = c.evaluate(half, noise=eps[sl], minibatch=rows[sl], want_fvalues=True)
        # f carries the entropy constant log(number_samples): log 128 here, log 256 in the full run
        parts.append(r["f"] + float(np.log(N) - np.log(half)))
        gsum += r["grads"] * half
    f_halves = torch.cat(parts)
    assert rel_err(f_halves.cpu().numpy(), f_rows.cpu().numpy()) <= 1e-6
    # BlackBox weights every row's score term with f, which contains that constant: only the pathwise part of the
    # gradient is shard-additive for a fixed N, so compare through the pathwise program on the same draws
    cp = engine.compile_model(model, model.posterior_model, "pathwise")
    full_p = cp.evaluate(N, noise=eps, minibatch=rows)
    gp, loss_p = full_p["grads"].clone(), float(full_p["loss"])     # views of the engine's output block: copy now
    gsum.zero_()
    lsum = 0.0
    for h in range(2):
        sl = slice(h * half, (h + 1) * half)
        r = cp.evaluate(half, noise=eps[sl], minibatch=rows[sl])
        gsum += r["grads"] * half
        lsum += (float(r["loss"]) + float(np.log(half))) * half        # loss = -mean f, f contains +log(n)
    assert abs(lsum / N - (loss_p + float(np.log(N)))) <= 1e-5 * abs(loss_p)
    scale = float(gp.abs().max())
    assert float((gsum / N - gp).abs().max()) <= 1e-5 * scale
    # every sum over rows is taken in a fixed order (one partial per slice of the rows, slices added in order by ONE
    # launch at the end; no float atomics): the whole output block — loss, counts, 668 948 gradients — is identical bit for
    # bit call after call, also with the weight-gradient launches running on the side stream
    for _ in range(3):
        again = cp.evaluate(N, noise=eps, minibatch=rows)
        assert float(again["loss"]) == loss_p
        assert torch.equal(again["grads"], gp)


def _custom_vae(api, enc_module, dec_module, dataset, batch_size, latent, prior_loc=None, prior_scale=None, bernoulli=False):
    import brancher_amd.functions as BF
    encoder, decoder = BF.BrancherFunction(enc_module), BF.BrancherFunction(dec_module)
    z = api.NormalVariable(np.zeros((latent,)) if prior_loc is None else prior_loc,
                           np.ones((latent,)) if prior_scale is None else prior_scale, name="z")
    out = api.DeterministicVariable(decoder(z), name="decoder_output")
    if bernoulli:
        x = api.BernulliVariable(logits=out["mean"], name="x")
    else:
        x = api.BinomialVariable(total_count=1, logits=out["mean"], name="x")
    model = api.ProbabilisticModel([x, z])
    Qx = api.EmpiricalVariable(dataset, batch_size=batch_size, name="x", is_observed=True)
    eo = api.DeterministicVariable(encoder(Qx), name="encoder_output")
    Qz = api.NormalVariable(eo["mean"], eo["sd"], name="z")
    model.set_posterior_model(api.ProbabilisticModel([Qx, Qz]))
    model.vae_modules = (enc_module, dec_module)
    return model


@pytest.mark.parametrize("partner", [1, 3, 5, 6])
def test_narrow_weight_gradient_is_bit_exact_beside_matrix_core_products_on_a_second_stream(partner):
    """The amortised path runs its weight gradients on a side stream beside the input gradients.  Round 4 found that a kernel
    with packed-f32 VALU instructions (outer_kernel, the narrow layers' weight gradient) returned wrong values in lanes 48-63
    while a bf16-MFMA kernel was resident on the same CUs from the other stream (10 of 24 launches, up to 4 % of a value;
    profiles/r4/x6_notes.txt section 4) — the translation unit is compiled without those instructions since.  Here: the
    encoder heads' weight gradient of cfg 5 (dW[4][512] over 25 600 rows) alone, then 16 times beside a partner product looping
    on another stream (mode 1 the f32-input MFMA kernel, 3 the exact-data bf16 x3 kernel, 5 / 6 the six-piece products):
    every launch must return the bits of the solo launch."""
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    g = torch.Generator(device="cpu").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    K, M, N = 25600, 4, 512
    A, B = rnd(K, M), rnd(K, N)
    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()

    def outer(out):
        out.zero_()
        native.check(lib.bsvi_debug_gemm(2, ptr(A), ptr(B), ptr(out), None, M, N, K, M, N, N, None, 0, 0, 0.0, 0, C.c_void_p(s_side.cuda_stream)))

    Mg, Ng, Kg = 25600, 256, 512
    Ag, Bnn, Bnt, Yg, bias = rnd(Mg, Kg), rnd(Kg, Ng), rnd(Ng, Kg), rnd(Mg, Ng), rnd(Ng)
    Xexact = (torch.rand(Mg, Kg, generator=g) > 0.5).float().to(dev)
    Cg = torch.zeros(Mg, Ng, device=dev)

    def product():
        st = C.c_void_p(s_main.cuda_stream)
        if partner == 1:
            native.check(lib.bsvi_debug_gemm(1, ptr(Ag), ptr(Bnn), ptr(Cg), None, Mg, Ng, Kg, Kg, Ng, Ng, ptr(Yg), Ng, 1, 0.0, 0, st))
        elif partner == 6:
            native.check(lib.bsvi_debug_gemm(6, ptr(Ag), ptr(Bnn), ptr(Cg), None, Mg, Ng, Kg, Kg, Ng, Ng, ptr(Yg), Ng, 1, 0.0, 0, st))
        elif partner == 5:
            native.check(lib.bsvi_debug_gemm(5, ptr(Ag), ptr(Bnt), ptr(Cg), None, Mg, Ng, Kg, Kg, Kg, Ng, ptr(bias), 0, 1, 0.0, 0, st))
        else:
            native.check(lib.bsvi_debug_gemm(3, ptr(Xexact), ptr(Bnt), ptr(Cg), None, Mg, Ng, Kg, Kg, Kg, Ng, ptr(bias), 0, 1, 0.0, 0, st))

    torch.cuda.synchronize()
    solo = torch.zeros(M, N, device=dev)
    with torch.cuda.stream(s_side):
        outer(solo)
    torch.cuda.synchronize()
    ref = A.double().T @ B.double()
    assert float((solo.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    outs = [torch.zeros(M, N, device=dev) for _ in range(16)]
    for o in outs:
        with torch.cuda.stream(s_main):
            product()
            product()
        with torch.cuda.stream(s_side):
            outer(o)
    torch.cuda.synchronize()
    bad = [int((o != solo).sum()) for o in outs]
    assert sum(bad) == 0, bad


@pytest.mark.parametrize("x6_modes", ["1", "3"])
def test_deep_networks_on_the_six_piece_products_match_oracle(x6_modes, monkeypatch):
    """Five wide layers at 384 rows: every one of them multiplies on x6gemm_kernel (forward; with BSVI_X6_MODES=3 the input
    gradients too), with widths that leave tails in every tile dimension (68, 132, 200, 72 are multiples of 4 only), and their
    ten weight splits + the data layer's do not fit the head launch's table of eight — the separate split launches serve them.
    Against the fp64 oracle, both estimators; the engine's gradients are bit-identical call to call."""
    import torch.nn as nn
    from brancher_amd import engine, workloads as W
    from oracle.vae_oracle import VaeOracle
    monkeypatch.setenv("BSVI_X6_MODES", x6_modes)
    rng = np.random.RandomState(77)
    P, latent, DS, B, N = 96, 3, 120, 48, 8
    torch.manual_seed(5)

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2, self.l3 = nn.Linear(P, 132), nn.Linear(132, 68), nn.Linear(68, 200)
            self.mean, self.sd, self.sp = nn.Linear(200, latent), nn.Linear(200, latent), nn.Softplus()

        def forward(self, x):
            h = torch.relu(self.l3(torch.relu(self.l2(torch.relu(self.l1(x.squeeze(-1)))))))
            return {"mean": self.mean(h), "sd": self.sp(self.sd(h)) + 0.05}

    class Dec(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2, self.l3, self.out = nn.Linear(latent, 72), nn.Linear(72, 200), nn.Linear(200, 132), nn.Linear(132, P)

        def forward(self, z):
            return {"mean": self.out(torch.relu(self.l3(torch.relu(self.l2(torch.relu(self.l1(z)))))))}

    enc, dec = Enc(), Dec()
    data = (rng.rand(DS, P, 1) > 0.5).astype("int32")
    rows = np.stack([rng.choice(DS, B, replace=False) for _ in range(N)])
    eps = rng.randn(N, B, latent).astype(np.float32)
    for estimator in ("pathwise", "blackbox"):
        model = _custom_vae(W.native_api(), enc, dec, data, B, latent)
        ref = VaeOracle(model, dtype=torch.float64).loss_and_grads(rows, eps, estimator)
        c = engine.compile_model(model, model.posterior_model, estimator)
        res = c.evaluate(N, noise=eps, minibatch=rows)
        first = c.out.clone()
        assert abs(float(res["loss"]) - ref["loss"]) <= TOL * max(abs(ref["loss"]), 1.0), (estimator, float(res["loss"]), ref["loss"])
        grads = _module_view(c, c.named_grads())
        # the suite's bound (conftest.yardstick_grad_check): as close to the double-precision oracle as the reference's own arithmetic —
        # the same oracle in single precision — is (x4), or within 1e-5 of the largest gradient
        ref32 = VaeOracle(_custom_vae(W.native_api(), enc, dec, data, B, latent), dtype=torch.float32).loss_and_grads(rows, eps, estimator)
        yardstick_grad_check(grads, ref["grads"], ref32["grads"])
        for _ in range(3):
            c.evaluate(N, noise=eps, minibatch=rows)
            assert torch.equal(c.out, first)
    # the two settings are different launch sequences: their gradients agree to rounding, not to the bit
    monkeypatch.setenv("BSVI_X6_MODES", "1" if x6_modes == "3" else "3")
    c.evaluate(N, noise=eps, minibatch=rows)
    assert not torch.equal(c.out, first)
    assert float((c.out - first).abs().max()) <= 2e-5 * float(first.abs().max())


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BSVI_TEST_SEEDS", "20"))))
def test_random_architectures_match_oracle(seed):
    """less-travelled shapes: heads straight off the data rows (gathered narrow layers), single-layer decoders, no
    biases, latent sizes 1..9, batch == dataset, widths around the tile edges — against the fp64 oracle, both estimators"""
    import torch.nn as nn
    from brancher_amd import engine, workloads as W
    from oracle.vae_oracle import VaeOracle
    rng = np.random.RandomState(100 + seed)
    P = int(rng.choice([5, 17, 64, 129, 200]))
    latent = int(rng.choice([1, 2, 3, 8, 9]))
    DS = int(rng.choice([7, 33, 64]))
    B = DS if seed % 4 == 0 else int(rng.randint(1, DS + 1))
    N = int(rng.choice([1, 3, 10]))
    enc_hidden = [int(rng.choice([4, 9, 65, 130])) for _ in range(int(rng.randint(0, 3)))]
    dec_hidden = [int(rng.choice([3, 8, 66, 128])) for _ in range(int(rng.randint(0, 3)))]
    bias = bool(rng.randint(0, 2))
    torch.manual_seed(seed)

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            dims = [P] + enc_hidden
            self.trunk = nn.ModuleList([nn.Linear(a, b, bias=bias) for a, b in zip(dims[:-1], dims[1:])])
            self.mean = nn.Linear(dims[-1], latent, bias=bias)
            self.sd = nn.Linear(dims[-1], latent)
            self.sp = nn.Softplus()

        def forward(self, x):
            h = x.squeeze(-1)
            for l in self.trunk:
                h = torch.relu(l(h))
            return {"mean": self.mean(h), "sd": self.sp(self.sd(h)) + 0.05}

    class Dec(nn.Module):
        def __init__(self):
            super().__init__()
            dims = [latent] + dec_hidden
            self.trunk = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
            self.out = nn.Linear(dims[-1], P, bias=bias)

        def forward(self, z):
            h = z
            for l in self.trunk:
                h = torch.relu(l(h))
            return {"mean": self.out(h)}

    enc, dec = Enc(), Dec()
    data = (rng.rand(DS, P, 1) > 0.5).astype("int32")
    prior_loc, prior_scale = rng.randn(latent) * 0.3, 0.5 + rng.rand(latent)
    build = lambda: _custom_vae(W.native_api(), enc, dec, data, B, latent, prior_loc, prior_scale, bernoulli=seed % 2 == 1)
    rows = np.stack([rng.choice(DS, B, replace=False) for _ in range(N)])
    eps = rng.randn(N, B, latent).astype(np.float32)
    for estimator in ("pathwise", "blackbox"):
        model = build()
        ref = VaeOracle(model, dtype=torch.float64).loss_and_grads(rows, eps, estimator)
        c = engine.compile_model(model, model.posterior_model, estimator)
        res = c.evaluate(N, noise=eps, minibatch=rows)
        assert abs(float(res["loss"]) - ref["loss"]) <= TOL * max(abs(ref["loss"]), 1.0), (estimator, float(res["loss"]), ref["loss"])
        grads = _module_view(c, c.named_grads())
        ref32 = VaeOracle(build(), dtype=torch.float32).loss_and_grads(rows, eps, estimator)
        yardstick_grad_check(grads, ref["grads"], ref32["grads"])      # (err <= max(4 |oracle_fp32 - oracle_fp64|, 1e-5 scale))


def test_the_fused_likelihood_epilogue_equals_the_separate_launch():
    """Round 6: the Bernoulli likelihood in the epilogue of the product that makes the logits (x6_epilogue, LIK: the logits never reach
    memory, amort_lik is not launched) against the separate launch (BSVI_AMORT_FUSE_LIK=0) at cfg 5's layer widths and 512 rows: the same
    per-row f, log q and gradients to rounding (the row sums of the log-likelihood are taken in another order), each bit-identical call
    to call; both estimators."""
    import os
    from brancher_amd import engine, workloads as W
    kw = dict(dataset_size=300, batch_size=64, n_features=784, latent_size=2, hidden1=512, hidden2=256, seed=3)
    for estimator in ("pathwise", "blackbox"):
        out = {}
        for fused in ("1", "0"):
            os.environ["BSVI_AMORT_FUSE_LIK"] = fused
            try:
                c = engine.compile_model(W.build_vae(W.native_api(), **kw), None, estimator)
                assert c.data_path() == "bf16x3"
                a = c.evaluate(8, seed=4, offset=2, want_fvalues=True)
                first = c.out.clone()
                fa = a["f"].clone()
                b = c.evaluate(8, seed=4, offset=2, want_fvalues=True)
                assert torch.equal(c.out, first) and torch.equal(b["f"], fa)
                out[fused] = (first, fa)
            finally:
                del os.environ["BSVI_AMORT_FUSE_LIK"]
        (g1, f1), (g0, f0) = out["1"], out["0"]
        assert not torch.equal(f1, f0)                                  # (another launch sequence: another summation order)
        assert float((f1 - f0).abs().max()) <= 2e-6 * float(f0.abs().max())
        assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())
