"""round 5: sha256 of the code objects hiprtc makes for BASELINE config 1's loop kernels (variant 0, + draw wave, + exchange) in THIS
process — run plain, after the GPU is initialised, and under rocprofv3 (tools/r5/jit_under_profiler.sh) to see whether any of these
contexts changes the code for an unchanged cache key.  python3 tools/r5/jit_hashes.py <tag> [init_gpu]"""
import hashlib
import json
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["BSVI_JIT_CACHE"] = "0"
if len(sys.argv) > 2 and sys.argv[2] == "init_gpu":
    import torch
    torch.zeros(4, device="cuda").sum().item()
from brancher_amd import lowering, native, workloads as W
m = W.build_readme_ar(W.native_api(), T=20)
src0 = native.specialised_source(lowering.lower(m, m.posterior_model, "pathwise"), 0)
variants = dict(plain=src0, draw_wave="#define SPEC_WITH_DRAW_WAVE 1\n" + src0,
                exchange="#define SPEC_WITH_EXCHANGE 1\n#define SPEC_WITH_DRAW_WAVE 1\n" + src0)
out = dict(tag=sys.argv[1] if len(sys.argv) > 1 else "")
d = os.environ.get("JIT_HASHES_KEEP") or tempfile.mkdtemp()
os.makedirs(d, exist_ok=True)
for name, src in variants.items():
    path = os.path.join(d, name + ".co")
    os.environ["BSVI_JIT_DUMP"] = path
    native.jit_compile(src)
    out[name] = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
out["env"] = sorted(k for k in os.environ if any(t in k for t in ("ROCP", "HSA", "HIP", "COMGR", "AMD", "LD_PRELOAD", "ROCM", "LLVM")))
print(json.dumps(out))
