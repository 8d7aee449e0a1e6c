"""
ORACLE TOOLING — records what the REAL reference's frame-level samplers return for the amortised model of BASELINE
config 5 (container only): `ProbabilisticModel.get_sample` (variables.py:751-757, the posterior-predictive step of
examples/VAE_playground.py:90-103) and `get_posterior_sample` (variables.py:796-812).  The draws themselves are random;
the fixture holds the STRUCTURE (columns, raw sample shapes) and, for a given latent value, the decoder output the
reference computes — tests/golden/frames/vae_frames.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vae_frames.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import reference_api, OUT  # noqa: E402

KW = dict(dataset_size=20, batch_size=5, n_features=12, hidden1=8, hidden2=6)


def shape_of(v):
    if isinstance(v, dict):
        return {k: shape_of(x) for k, x in v.items()}
    return list(v.shape) if hasattr(v, "shape") else "scalar"


if __name__ == "__main__":
    import torch
    api = reference_api()
    from brancher import standard_variables as sv
    api.BinomialVariable = sv.BinomialVariable
    import brancher_amd.workloads as W
    model = W.build_vae(api, **KW)
    np.random.seed(3)
    torch.manual_seed(3)
    z = model.get_variable("z")
    z_value = np.array([0.3, -1.2], dtype=np.float32)
    meta = dict(builder="build_vae", kwargs=KW, reference="LucaAmbrogioni/Brancher @ /root/reference")
    raw = model._get_sample(2)
    meta["get_sample_raw"] = {v.name: shape_of(s) for v, s in raw.items()}
    frame = model.get_sample(2)
    meta["get_sample_columns"] = list(frame.columns)
    given = model.get_sample(1, input_values={z: z_value})
    meta["get_sample_given_columns"] = list(given.columns)
    post = model._get_posterior_sample(3)
    meta["get_posterior_sample_raw"] = {v.name: shape_of(s) for v, s in post.items()}
    meta["get_posterior_sample_columns"] = list(model.get_posterior_sample(3).columns)
    out = dict(meta=np.array(json.dumps(meta)), z_value=z_value,
               decoder_mean_given_z=np.asarray(given["decoder_output"].values[0]["mean"], dtype=np.float32),
               z_given_cell=np.asarray(given["z"].values[0], dtype=np.float32),
               x_given_cell=np.asarray(given["x"].values[0], dtype=np.float32))
    os.makedirs(os.path.join(OUT, "frames"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "frames", "vae_frames.npz"), **out)
    print(json.dumps(meta, indent=1))
    print({k: v.shape for k, v in out.items() if k != "meta"})
