"""Node arithmetic of the kernel (csrc/dist_math.h) against torch.distributions on CPU —
value and every partial derivative, per distribution — through the bsvi_debug_math hook."""
import ctypes as C

import numpy as np
import pytest
import torch
from torch import distributions as td

from brancher_amd import native
from brancher_amd import distributions as D

pytestmark = pytest.mark.gpu


def run(fn, dist, x, p0, p1):
    lib = native.load()
    dev = torch.device("cuda:0")
    n = len(x)
    xs, a, b = (torch.tensor(np.asarray(v, dtype=np.float32)).to(dev) for v in (x, p0, p1))
    out = torch.zeros(4 * n, device=dev)
    native.check(lib.bsvi_debug_math(fn, dist, C.c_void_p(xs.data_ptr()), C.c_void_p(a.data_ptr()),
                                     C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), n, None))
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(4, n)


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.all(np.abs(a - b) <= tol * (1.0 + np.abs(b))), np.abs(a - b).max()


def test_special_functions():
    x = np.concatenate([np.linspace(0.05, 3, 40), np.linspace(3, 60, 40)]).astype(np.float32)
    t = torch.tensor(x)
    close(run(0, 0, x, x, x)[0], torch.digamma(t).numpy())
    close(run(1, 0, x, x, x)[0], torch.polygamma(1, t).numpy())
    close(run(6, 0, x, x, x)[0], torch.lgamma(t).numpy())


def test_dirichlet_grad_all_branches():
    rng = np.random.RandomState(0)
    alpha = np.exp(rng.uniform(np.log(0.1), np.log(40), 4000)).astype(np.float32)
    beta = np.exp(rng.uniform(np.log(0.1), np.log(40), 4000)).astype(np.float32)
    x = rng.beta(alpha, beta).astype(np.float32).clip(1e-6, 1 - 1e-6)
    total = alpha + beta
    ref = torch._dirichlet_grad(torch.tensor(x), torch.tensor(alpha), torch.tensor(total)).numpy()
    got = run(2, 0, x, alpha, total)[0]
    close(got, ref, 5e-5)


CASES = [
    (D.DIST_NORMAL, lambda a, b: td.Normal(a, b), "real"),
    (D.DIST_LOGNORMAL, lambda a, b: td.LogNormal(a, b), "pos"),
    (D.DIST_CAUCHY, lambda a, b: td.Cauchy(a, b), "real"),
    (D.DIST_LAPLACE, lambda a, b: td.Laplace(a, b), "real"),
    (D.DIST_BETA, lambda a, b: td.Beta(a, b), "unit"),
]


@pytest.mark.parametrize("dist,make,support", CASES)
def test_logp_entropy_and_gradients(dist, make, support):
    rng = np.random.RandomState(dist)
    n = 512
    if dist == D.DIST_BETA:
        p0 = np.exp(rng.uniform(-1.5, 2.5, n)); p1 = np.exp(rng.uniform(-1.5, 2.5, n))
        p0[:8] = 1.0; p1[4:12] = 1.0       # torch.xlogy masks the (alpha-1)==0 terms
    else:
        p0 = rng.normal(0, 2, n); p1 = np.exp(rng.uniform(-2, 1.5, n))
    x = {"real": rng.normal(0, 3, n), "pos": np.exp(rng.normal(0, 1, n)), "unit": rng.uniform(0.02, 0.98, n)}[support]
    x, p0, p1 = (v.astype(np.float32) for v in (x, p0, p1))
    tx, ta, tb = (torch.tensor(v, requires_grad=True) for v in (x, p0, p1))
    lp = make(ta, tb).log_prob(tx)
    lp.sum().backward()
    got = run(3, dist, x, p0, p1)
    close(got[0], lp.detach().numpy())
    close(got[1], tx.grad.numpy()); close(got[2], ta.grad.numpy()); close(got[3], tb.grad.numpy())
    ta.grad = None; tb.grad = None
    H = make(ta, tb).entropy()
    H.sum().backward()
    got = run(4, dist, x, p0, p1)
    close(got[0], H.detach().numpy())
    close(got[2], np.zeros(n) if ta.grad is None else ta.grad.numpy())
    close(got[3], tb.grad.numpy())


def test_discrete_logp():
    rng = np.random.RandomState(3)
    n = 256
    logits = rng.normal(0, 3, n).astype(np.float32)
    total = rng.randint(1, 12, n).astype(np.float32)
    k = np.floor(rng.uniform(0, 1, n) * (total + 1)).clip(0, total).astype(np.float32)
    tl = torch.tensor(logits, requires_grad=True)
    lp = td.Binomial(torch.tensor(total), logits=tl).log_prob(torch.tensor(k))
    lp.sum().backward()
    got = run(3, D.DIST_BINOMIAL, k, total, logits)
    close(got[0], lp.detach().numpy()); close(got[3], tl.grad.numpy())
    xb = (rng.uniform(0, 1, n) < 0.5).astype(np.float32)
    tl = torch.tensor(logits, requires_grad=True)
    lp = td.Bernoulli(logits=tl).log_prob(torch.tensor(xb))
    lp.sum().backward()
    got = run(3, D.DIST_BERNOULLI, xb, logits, logits)
    close(got[0], lp.detach().numpy()); close(got[2], tl.grad.numpy())
    tl.grad = None
    H = td.Bernoulli(logits=tl).entropy()
    H.sum().backward()
    got = run(4, D.DIST_BERNOULLI, xb, logits, logits)
    close(got[0], H.detach().numpy()); close(got[2], tl.grad.numpy())
