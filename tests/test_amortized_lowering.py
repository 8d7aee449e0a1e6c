"""Host side of the amortised path (brancher_amd/amortized.py) on CPU: the torch.fx lowering of encoder / decoder
modules, the graph pattern match, the parameter layout and the C-ABI descriptors.  No compute calls."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from brancher_amd import amortized, workloads as W
from brancher_amd import functions as BF
from brancher_amd.lowering import LoweringError


def _link(module):
    return BF.BrancherFunction(module).fn


def test_trace_modules_that_override_call():
    """the reference's examples define networks by overriding __call__ and use activation MODULES
    (examples/VAE_playground.py:27-62); torch.fx must see through both"""
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.act, self.sp = nn.Linear(6, 4), nn.Linear(4, 3), nn.ReLU(), nn.Softplus()

        def __call__(self, x):
            return {"s": self.sp(self.b(self.act(self.a(x.squeeze())))) + 0.25}

    layers, outs = amortized.trace_network(_link(Net()))
    assert [(l.n_in, l.n_out, l.activation, l.post_add) for l in layers] == [(6, 4, 1, 0.0), (4, 3, 2, 0.25)] and outs == {"s": 2}


def test_trace_workload_modules():
    enc, dec = W.vae_modules(n_features=12, latent_size=2, hidden1=8, hidden2=6, seed=0)
    layers, outs = amortized.trace_network(_link(enc))
    assert [(l.in_value, l.out_value, l.n_in, l.n_out, l.activation) for l in layers] == \
        [(0, 1, 12, 6, amortized.ACT_RELU), (1, 2, 6, 8, amortized.ACT_RELU), (2, 3, 8, 2, amortized.ACT_NONE),
         (2, 4, 8, 2, amortized.ACT_SOFTPLUS)]
    assert layers[3].post_add == pytest.approx(0.1) and layers[2].post_add == 0.0
    assert outs == {"mean": 3, "sd": 4}
    layers, outs = amortized.trace_network(_link(dec))
    assert [(l.n_in, l.n_out, l.activation) for l in layers] == [(2, 8, 1), (8, 6, 1), (6, 12, 0)] and outs == {"mean": 3}


def test_trace_sequential_and_functional_forms():
    import torch.nn.functional as F

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Linear(5, 7), nn.ReLU(), nn.Linear(7, 3, bias=False))

        def forward(self, x):
            return F.softplus(self.body(x.flatten(1))) + 0.5

    layers, outs = amortized.trace_network(_link(Net()))
    assert [(l.n_in, l.n_out, l.activation, l.post_add) for l in layers] == [(5, 7, 1, 0.0), (7, 3, 2, 0.5)]
    assert layers[1].bias is None and outs == {None: 2}


@pytest.mark.parametrize("bad", ["tanh", "shared_preactivation", "relu_after_shift", "two_inputs"])
def test_unsupported_networks_fail_loudly(bad):
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.l = nn.Linear(4, 4)

        def forward(self, x, y=None):
            h = self.l(x)
            if bad == "tanh":
                return torch.tanh(h)
            if bad == "shared_preactivation":
                return {"a": torch.relu(h), "b": h}
            if bad == "relu_after_shift":
                return torch.relu(h + 1.0)
            return h + y

    with pytest.raises(LoweringError):
        amortized.trace_network(_link(Net()))


def test_lowering_of_the_vae_graph():
    model = W.build_vae(W.native_api(), dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6)
    p = amortized.lower_amortized(model, model.posterior_model, "blackbox")
    assert (p.n_features, p.latent_dim, p.dataset_size, p.batch_size) == (12, 2, 20, 5)
    assert np.array_equal(p.prior_loc, [0, 0]) and np.allclose(p.prior_scale, [1, 1])
    assert p.dataset.shape == (20, 12) and set(np.unique(p.dataset)) <= {0.0, 1.0}
    # one flat buffer: encoder tensors (optimizer group 0) then decoder tensors (group 1), no gaps, torch layout
    enc, dec = model.vae_modules
    n_tensors = sum(t.numel() for m in (enc, dec) for t in m.parameters())
    assert n_tensors <= p.n_params <= n_tensors + 3            # at most one alignment gap between the two networks
    cover = np.zeros(p.n_params, dtype=int)
    for par, off, size, group in p.parameters:
        cover[off:off + size] += 1
        assert np.all(p.param_group[off:off + size] == group)
    assert cover.max() == 1 and np.array_equal(cover, p.param_active)   # padding elements are inactive
    l1 = p.enc_layers[0]
    assert np.array_equal(l1.weight.numpy(), enc.l1.weight.detach().numpy()) and l1.weight.shape == (6, 12)
    # every weight matrix whose size allows it starts 16-byte aligned
    for l in p.enc_layers + p.dec_layers:
        if (l.n_in * l.n_out) % 4 == 0:
            assert l.weight_off % 4 == 0


def test_sibling_heads_are_merged_into_one_layer():
    model = W.build_vae(W.native_api(), dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6)
    p = amortized.lower_amortized(model, model.posterior_model, "pathwise")
    assert [(l.n_in, l.n_out) for l in p.enc_layers] == [(12, 6), (6, 8), (8, 4)]
    head = p.enc_layers[-1]
    assert (head.split_col, head.activation, head.activation2, head.post_add2) == (2, amortized.ACT_NONE, amortized.ACT_SOFTPLUS,
                                                                                   pytest.approx(0.1))
    assert (p.enc_loc_value, p.enc_loc_col, p.enc_scale_value, p.enc_scale_col) == (head.out_value, 0, head.out_value, 2)
    # the two heads' tensors are adjacent, in head order: the merged layer reads them as one [4][8] matrix / [4] bias
    first, second = head.parts
    off = {id(par): o for par, o, _, _ in p.parameters}
    assert off[id(second.weight)] == off[id(first.weight)] + first.weight.size == head.weight_off + first.weight.size
    assert off[id(second.bias)] == off[id(first.bias)] + first.bias.size


def test_graphs_outside_the_pattern_are_rejected():
    api = W.native_api()
    enc, dec = W.vae_modules(12, 2, 8, 6)
    data = W.vae_data(20, 12)
    z = api.NormalVariable(np.zeros((2,)), np.ones((2,)), name="z")
    out = api.DeterministicVariable(BF.BrancherFunction(dec)(z), name="decoder_output")
    x = api.NormalVariable(out["mean"], 1.0, name="x", learnable=True)       # a Normal likelihood with a LEARNABLE scale
    model = api.ProbabilisticModel([x, z])
    Qx = api.EmpiricalVariable(data, batch_size=5, name="x", is_observed=True)
    eo = api.DeterministicVariable(BF.BrancherFunction(enc)(Qx), name="encoder_output")
    Qz = api.NormalVariable(eo["mean"], eo["sd"], name="z")
    model.set_posterior_model(api.ProbabilisticModel([Qx, Qz]))
    # (round 4: served — the scale's raw value joins the joint model's group behind the networks' tensors, ABI 10)
    p = amortized.lower_amortized(model, model.posterior_model, "pathwise")
    names = {par.name: (off, size, group) for par, off, size, group in p.parameters}
    assert names["x_scale"] == (p.lik_scale_off, 1, 1) and p.lik_scale_size == 1 and p.n_params == p.lik_scale_off + 1
    assert np.allclose(p.likelihood_scale, 1.0, atol=1e-6)                # the initial value, behind softplus
    # an ARITHMETIC expression of a decoder output as the scale is still outside the pattern (a second head is inside: below)
    x2 = api.NormalVariable(out["mean"], BF.exp(out["mean"]), name="x")
    model2 = api.ProbabilisticModel([x2, z])
    model2.set_posterior_model(api.ProbabilisticModel([Qx, Qz]))
    with pytest.raises(LoweringError):
        amortized.lower_amortized(model2, model2.posterior_model, "pathwise")
    # (round 4) NormalVariable(decoder(z)["mean"], decoder(z)["sd"]): the scale is a second head of the decoder
    m3 = W.build_vae(api, dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6, likelihood="normal", likelihood_scale="decoder")
    p3 = amortized.lower_amortized(m3, m3.posterior_model, "blackbox")
    heads = [l for l in p3.dec_layers if l.out_value in (p3.dec_logits_value, p3.dec_scale_value)]
    assert p3.dec_scale_key == "sd" and p3.dec_scale_value not in (0, p3.dec_logits_value) and len(heads) == 2
    assert heads[0].in_value == heads[1].in_value and p3.lik_scale_size == 0
    with pytest.raises(LoweringError):
        amortized.lower_amortized(W.build_vae(api, dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6),
                                  None if False else W.build_vae(api, dataset_size=20, batch_size=5, n_features=12,
                                                                  hidden1=8, hidden2=6).posterior_model, "taylor1")


def test_widened_pattern_normal_likelihood_and_learnable_prior():
    """Normal(decoder(z), constant scale) likelihood; NormalVariable(..., learnable=True) prior: its two roots join the
    decoder's optimizer group behind the networks' tensors (inference.py:77-88)"""
    api = W.native_api()
    kw = dict(dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6)
    p = amortized.lower_amortized(*(lambda m: (m, m.posterior_model))(W.build_vae(api, likelihood="normal", likelihood_scale=0.7, **kw)))
    assert p.likelihood == "normal" and p.likelihood_scale.shape == (12,) and np.allclose(p.likelihood_scale, 0.7, rtol=1e-6)
    assert p.prior_loc_off == p.prior_scale_off == amortized.NO_BIAS
    assert p.dataset.dtype == np.float32 and len(np.unique(p.dataset)) > 2
    per_feature = [0.4 + 0.01 * j for j in range(12)]
    m = W.build_vae(api, likelihood="normal", likelihood_scale=per_feature, learnable_prior=True, latent_size=3, **kw)
    p = amortized.lower_amortized(m, m.posterior_model, "blackbox")
    assert np.allclose(p.likelihood_scale, per_feature, rtol=1e-6)
    names = {par.name: (off, size, group) for par, off, size, group in p.parameters}
    assert names["z_loc"] == (p.prior_loc_off, 3, 1) and names["z_scale"] == (p.prior_scale_off, 3, 1)
    assert p.prior_scale_off == p.prior_loc_off + 3 and p.n_params == p.prior_scale_off + 3
    assert np.all(p.param_active[p.prior_loc_off:] == 1) and np.all(p.param_group[p.prior_loc_off:] == 1)
    raw = [par for par, *_ in p.parameters if par.name == "z_scale"][0].numpy().reshape(-1)
    assert np.allclose(np.log1p(np.exp(raw)), 1.0, atol=1e-6)             # stored behind the softplus of its range
    with pytest.raises(LoweringError):                                    # a scale that is neither one number nor one per feature
        bad = W.build_vae(api, likelihood="normal", likelihood_scale=[0.5, 0.6, 0.7], **kw)
        amortized.lower_amortized(bad, bad.posterior_model)


def test_c_abi_structs_match_the_header():
    import ctypes as C
    import re
    from brancher_amd import native
    from conftest import ROOT
    import os
    header = open(os.path.join(ROOT, "include", "bsvi.h")).read()
    assert C.sizeof(native.MlpLayer) == 48
    body = re.search(r"typedef struct bsvi_mlp_layer \{(.*?)\} bsvi_mlp_layer;", header, re.S).group(1)
    names = re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert [n for n, _ in native.MlpLayer._fields_] == names
    body = re.search(r"typedef struct bsvi_amort_args \{(.*?)\} bsvi_amort_args;", header, re.S).group(1)
    names = re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert [n for n, _ in native.AmortArgs._fields_] == names
    body = re.search(r"typedef struct bsvi_amort_desc \{(.*?)\} bsvi_amort_desc;", header, re.S).group(1)
    names = re.findall(r"(\w+)\s*(?:,|;)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert [n for n, _ in native.AmortDesc._fields_] == names
