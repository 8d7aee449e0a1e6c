"""Writes tests/c_abi/readme_ar_T20_N300.blob: everything a C program needs to drive the README autoregressive model
(BASELINE config 1) through the C ABI with no Python in the process — the lowered program tables, the initial parameters,
the observations, the noise of the reference fixture tests/golden/readme_ar_T20_N300.npz in the kernel's layout
([noise row][sample]) and the REFERENCE's loss and gradients (Pathwise) for that noise.

    python tests/c_abi/make_blob.py          (CPU only; rerun when the lowering or the fixture changes)

Layout (little endian): magic "BSVIBLOB", u32 version, u32 abi_version, then 12 u32 scalars
(n_params n_consts n_obs n_slots n_noise n_uniform n_uniform_grad n_records n_code estimator n_samples reserved), then the
arrays in this order, each as u64 byte count + bytes padded to 8: uniform, records, code, consts, param_uniform_ptr,
param_uniform_idx, params, obs, noise, reference {loss, grads[n_params]} as f32."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import Golden                                  # noqa: E402
from brancher_amd import engine, lowering, native            # noqa: E402

NAME = "readme_ar_T20_N300"


def main():
    g = Golden(NAME)
    prog = lowering.lower(g.build(), None, "pathwise")
    d, keep = native.program_desc(prog)
    noise = engine.noise_from_named(prog, g.noise, g.N)                       # [n_noise][N]
    grads = np.zeros(prog.n_params, dtype=np.float32)
    ref = g.group("grad_pathwise/")
    for par, off, size, _ in prog.parameters:
        grads[off:off + size] = np.asarray(ref[par.name], dtype=np.float32).reshape(-1)
    reference = np.concatenate([[np.float32(g.data["loss_pathwise"])], grads]).astype(np.float32)
    arrays = [keep["uniform"], keep["records"], keep["code"], keep["consts"], keep["ptr"], keep["idx"],
              prog.initial_params().astype(np.float32), np.ascontiguousarray(prog.obs, dtype=np.float32), noise, reference]
    path = os.path.join(ROOT, "tests", "c_abi", NAME + ".blob")
    with open(path, "wb") as f:
        f.write(b"BSVIBLOB")
        f.write(struct.pack("<2I", 1, native.ABI_VERSION))
        f.write(struct.pack("<12I", d.n_params, d.n_consts, d.n_obs, d.n_slots, d.n_noise, d.n_uniform, d.n_uniform_grad,
                            d.n_records, d.n_code, d.estimator, g.N, 0))
        for a in arrays:
            raw = np.ascontiguousarray(a).tobytes()
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw + b"\0" * (-len(raw) % 8))
    print(path, os.path.getsize(path), "bytes; n_params", d.n_params, "n_noise", d.n_noise, "N", g.N)


if __name__ == "__main__":
    main()
