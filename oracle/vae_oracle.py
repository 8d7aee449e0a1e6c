"""
ORACLE — test infrastructure, not product code (see oracle/svi_oracle.py for the rules: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; brancher_amd/ never does).

CPU restatement of the reference's ELBO-gradient path for the amortised workload of BASELINE config 5
(`examples/VAE_playground.py:64-79`).  Like the reference it CALLS the user's torch modules (BrancherFunction ->
`self.fn(*args)`, `brancher/functions.py:36-39`) on the rows of the iteration and leaves the gradients to
autograd, so it shares the reference's numerics (torch.nn.Linear / ReLU / Softplus, torch.distributions) and
checks the torch.fx lowering of brancher_amd/amortized.py independently:

  x[n, b]   = dataset[rows[n, b]]                      EmpiricalDistribution._get_sample   distributions.py:436-457
  loc, sd   = encoder(x)                               DeterministicVariable / link        variables.py:436-449
  z         = loc + sd * eps                           Normal rsample                      distributions.py:111-124
  logits    = decoder(z)
  lp        = sum_j Binomial(1, logits).log_prob(x) + sum_d Normal(prior).log_prob(z)      variables.py:486-520
              (variants: sum_j Normal(decoder(z), constant scale).log_prob(x); a learnable prior Normal(loc, softplus(raw)))
  H         = sum_d Normal(loc, sd).entropy() + log(N)   (entropy of the minibatch variable, distributions.py:464-473:
              Categorical(ones(dataset.shape[0])) where the dataset has been tiled to N samples)   variables.py:156-162
  lq        = sum_d Normal(loc, sd).log_prob(z)
  pathwise  : loss = -mean_{n,b}(lp + H)                                                   gradient_estimators.py:41-44
  BlackBox  : loss = -mean_{n,b}(lq * stopgrad(lp + H) + lp + H)                           gradient_estimators.py:31-36
  loop      : first optimizer (posterior = encoder) every iteration, the model's (decoder) when iteration > 0
                                                                                           inference.py:95-108
Pinning: tests/golden/vae_*.npz, generated from the real reference by oracle/gen_golden_vae.py;
tests/test_oracle_golden.py checks this module against every one of them.
"""
import copy

import numpy as np
import torch
from torch import distributions as td


class VaeOracle:
    def __init__(self, model, dtype=torch.float32):
        from brancher_amd import amortized
        prog = amortized.lower_amortized(model, model.posterior_model, "pathwise")   # for the graph roles only
        enc_link, dec_link = prog.links
        self.dtype = dtype
        self.enc = copy.deepcopy(enc_link.module).to(dtype)
        self.dec = copy.deepcopy(dec_link.module).to(dtype)
        self.dataset = torch.from_numpy(np.asarray(prog.dataset)).to(dtype)          # [DS, P]
        self.prior_loc = torch.from_numpy(prog.prior_loc).to(dtype)
        self.prior_scale = torch.from_numpy(prog.prior_scale).to(dtype)
        self.B, self.Dz = prog.batch_size, prog.latent_dim
        # the widened pattern: Normal likelihood with a constant scale; a learnable prior (raw values: loc as stored, scale
        # behind softplus — standard_variables.py:57-68, geometric_ranges.py RightHalfLine)
        self.likelihood = prog.likelihood
        self.lik_scale = None if prog.likelihood_scale is None else torch.from_numpy(prog.likelihood_scale).to(dtype)
        # ... or learnable: NormalVariable(decoder value, scale, learnable=True) keeps softplus^-1(scale) as a root of the joint model
        self.lik_raw, self._lik_name = None, None
        self.scale_head_key = getattr(prog, "dec_scale_key", None)
        for par, off, size, group in prog.parameters:
            if getattr(prog, "lik_scale_size", 0) and off == prog.lik_scale_off:
                self.lik_raw = torch.nn.Parameter(torch.from_numpy(np.asarray(par.numpy(), dtype=np.float64)).to(dtype))
                self._lik_name = par.name
        self.prior_raw = {}
        for par, off, size, group in prog.parameters:
            if off in (prog.prior_loc_off, prog.prior_scale_off):
                self.prior_raw[par.name] = torch.nn.Parameter(torch.from_numpy(np.asarray(par.numpy(), dtype=np.float64)).to(dtype))
                if off == prog.prior_loc_off:
                    self._loc_name = par.name
                else:
                    self._scale_name = par.name
        self.prior_loc_learnable = prog.prior_loc_off != 0xFFFFFFFF
        self.prior_scale_learnable = prog.prior_scale_off != 0xFFFFFFFF

    def named_parameters(self):
        out = {}
        for tag, m in (("enc", self.enc), ("dec", self.dec)):
            for k, p in m.named_parameters():
                out["%s/%s" % (tag, k)] = p
        for k, p in self.prior_raw.items():
            out["prior/" + k] = p
        if self.lik_raw is not None:
            out["prior/" + self._lik_name] = self.lik_raw        # ("prior/": the fixtures' prefix for every learnable root of the joint model)
        return out

    def prior(self):
        loc = self.prior_raw[self._loc_name].reshape(-1) if self.prior_loc_learnable else self.prior_loc
        scale = torch.nn.functional.softplus(self.prior_raw[self._scale_name].reshape(-1)) if self.prior_scale_learnable \
            else self.prior_scale
        return loc, scale

    def terms(self, rows, eps):
        rows = torch.as_tensor(np.asarray(rows, dtype=np.int64))
        eps = torch.as_tensor(np.asarray(eps)).to(self.dtype).reshape(rows.shape + (self.Dz,))
        x = self.dataset[rows]                                   # [N, B, P]
        out = self.enc(x.unsqueeze(-1))                          # the reference hands rows over as [.., P, 1]
        loc, sd = out["mean"], out["sd"]
        z = loc + sd * eps
        dec_out = self.dec(z)
        logits = dec_out["mean"]
        ploc, pscale = self.prior()
        if self.likelihood == "normal":
            scale = self.lik_scale if self.lik_raw is None else torch.nn.functional.softplus(self.lik_raw.reshape(-1))
            if self.scale_head_key is not None:              # NormalVariable(decoder(z)["mean"], decoder(z)["sd"]): a second head
                scale = dec_out[self.scale_head_key]
            lik = td.Normal(logits, scale).log_prob(x).sum(-1)
        else:
            lik = td.Binomial(total_count=1, logits=logits).log_prob(x).sum(-1)
        lp = lik + td.Normal(ploc, pscale).log_prob(z).sum(-1)
        q = td.Normal(loc, sd)
        # the minibatch variable is part of the posterior and has an "analytic entropy": Categorical(ones(n)).entropy()
        # with n = dataset.shape[0], which for the sample-tiled dataset is the NUMBER OF SAMPLES, not the dataset
        # size (distributions.py:464-473) -> the constant log(N) enters every row's H (and BlackBox's score weight)
        H = q.entropy().sum(-1) + float(np.log(rows.shape[0]))
        return dict(lp=lp, H=H, lq=q.log_prob(z).sum(-1), z=z)

    def loss(self, rows, eps, estimator="pathwise"):
        t = self.terms(rows, eps)
        f = t["lp"] + t["H"]
        if callable(estimator):
            # a user-defined GradientEstimator (gradient_estimators.py:17-26) restated as its scalar g(f, log q) of the
            # [N, B] per-row values
            return -estimator(f, t["lq"]), t
        value = f if estimator == "pathwise" else t["lq"] * f.detach() + f
        return -value.mean(), t

    def loss_and_grads(self, rows, eps, estimator="pathwise"):
        params = self.named_parameters()
        for p in params.values():
            p.grad = None
        loss, t = self.loss(rows, eps, estimator)
        loss.backward()
        return dict(loss=float(loss.detach()), grads={k: p.grad.detach().numpy().copy() for k, p in params.items()},
                    f=(t["lp"] + t["H"]).detach().numpy(), lq=t["lq"].detach().numpy(), z=t["z"].detach().numpy())

    def train(self, iters, rows_seq, eps_seq, optimizer="Adam", **opt_kwargs):
        cls = getattr(torch.optim, optimizer)
        opts = [cls(self.enc.parameters(), **opt_kwargs),
                cls(list(self.dec.parameters()) + list(self.prior_raw.values()) + ([self.lik_raw] if self.lik_raw is not None else []),
                    **opt_kwargs)]
        losses = []
        for it in range(iters):
            loss, _ = self.loss(rows_seq[it], eps_seq[it], "pathwise")
            if torch.isfinite(loss.detach()).all().item():
                [o.zero_grad() for o in opts]
                loss.backward()
                opts[0].step()
                if it > 0:
                    opts[1].step()
            losses.append(float(loss.detach()))
        return np.array(losses)
