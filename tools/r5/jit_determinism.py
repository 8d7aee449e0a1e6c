"""round 5: is a code object a function of what the disk cache's key hashes?  The same generated translation units are compiled
(a) as the FIRST compilations of a process and (b) after N other programs of the same process (other models, a long BlackBox program
that carries -fno-slp-vectorize), and the code objects are compared byte for byte.  Needs no GPU (hiprtc cross-compiles).
python tools/r5/jit_determinism.py [n_other]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, %r)
from brancher_amd import lowering, native, workloads as W
api = W.native_api()
n_other = int(sys.argv[1]); out_dir = sys.argv[2]
def sources(builder, est, **kw):
    m = getattr(W, builder)(api, **kw)
    p = lowering.lower(m, m.posterior_model, est)
    return [native.specialised_source(p, v) for v in range(2)]
if n_other:
    others = [("build_readme_ar", "pathwise", dict(T=t)) for t in (3, 5, 7, 9, 11, 13)] + \
             [("build_readme_ar", "blackbox", dict(T=t)) for t in (4, 6, 8, 10)] + \
             [("build_beta_binomial", "pathwise", {}), ("build_beta_binomial", "blackbox", {}), ("build_readme_ar", "blackbox", dict(T=120))]
    k = 0
    for b, e, kw in others:
        for s in sources(b, e, **kw):
            if s and k < n_other:
                native.jit_compile(s); k += 1
res = {}
for tag, (b, e, kw) in dict(cfg1=("build_readme_ar", "pathwise", dict(T=20)), cfg1_bb=("build_readme_ar", "blackbox", dict(T=20)),
                            cfg2=("build_beta_binomial", "pathwise", {})).items():
    for v, s in enumerate(sources(b, e, **kw)):
        dump = os.path.join(out_dir, "%%s_v%%d.co" %% (tag, v))
        os.environ["BSVI_JIT_DUMP"] = dump
        native.jit_compile(s)
        res["%%s_v%%d" %% (tag, v)] = hashlib.sha256(open(dump, "rb").read()).hexdigest()
print(json.dumps(res))
''' % os.path.abspath(ROOT)

def run(n_other, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, BSVI_JIT_CACHE="0")
    out = subprocess.run([sys.executable, "-c", CHILD, str(n_other), out_dir], check=True, capture_output=True, text=True, env=env).stdout
    return json.loads(out.strip().splitlines()[-1])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
base = "/tmp/jit_determinism"
a, b, c = run(0, base + "/first"), run(n, base + "/late"), run(0, base + "/first_again")
for k in sorted(a):
    print("%-12s first == first again: %-5s   first == after %d others: %s" % (k, a[k] == c[k], n, a[k] == b[k]))
