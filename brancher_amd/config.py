"""
Global device selection.

Mirrors the reference's single global knob (`brancher/config.py:10-30`): one
module-level ``device`` plus ``set_device``.  Differences, all documented in
DESIGN.md:

* the default is the first HIP device when one is visible (the engine only runs on
  MI355X); ``'cpu'`` can still be selected so that graphs can be *built and lowered*
  on a machine without a GPU, but evaluating a model then fails loudly — there is no
  CPU execution path in this package.
* ``'hip'``, ``'hip:i'`` and ``'mi355x'`` are accepted as aliases of ``'cuda:i'``
  (torch-ROCm names HIP devices ``cuda``).
* an ``int`` is accepted (the reference calls ``.lower()`` before its ``isinstance(int)``
  branch, `config.py:14`, so that branch is dead there).
"""
import torch

default_device = "auto"
device = torch.device("cpu")


def _gpu_visible():
    # device_count() does not initialise the HIP runtime on this image
    try:
        return torch.cuda.device_count() > 0
    except Exception:  # pragma: no cover
        return False


def set_device(device_):
    global device
    if isinstance(device_, int):
        device = torch.device("cuda", device_)
        return device
    if isinstance(device_, torch.device):
        device = device_
        return device
    if not isinstance(device_, str):
        raise ValueError("Device is not recongnized")
    name = device_.lower()
    if name == "auto":
        device = torch.device("cuda:0") if _gpu_visible() else torch.device("cpu")
    elif name == "cpu":
        device = torch.device("cpu")
    elif name in ("gpu", "hip", "mi355x", "cuda"):
        assert _gpu_visible(), "GPU requested but not available"
        device = torch.device("cuda:0")
    elif name.startswith("cuda:") or name.startswith("hip:"):
        assert _gpu_visible(), "GPU requested but not available"
        device = torch.device("cuda:" + name.split(":", 1)[1])
    else:
        raise ValueError("Device is not recongnized")
    return device


def get_device():
    return device


set_device(default_device)
