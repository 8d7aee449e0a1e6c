"""Round 6 (missing 7): the wall time of the call the reference's user writes — README.md:71-75,

    inference.perform_inference(model, number_iterations=500, number_samples=300, optimizer="SGD", lr=0.001)

— model construction excluded, everything else included: optimizer objects, lowering, program creation, hiprtc (cold: an empty
code-object cache; cached: the second process), the 500-iteration launch, the copy of the loss curve to the host.  Beside it the same
process's second and third call (everything warm) and `compiled.train(500)` on the compiled object.
python3 tools/r6/perform_inference_readme.py            (runs itself twice as a child with a private cache directory)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    t_imp = time.perf_counter()
    from brancher_amd import inference, workloads as W, engine
    api = W.native_api()
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    out = dict(import_and_device_s=t_ready - t_imp)
    calls = []
    for i in range(3):
        model = W.build_readme_ar(api, T=20)          # a fresh model per call (what a script does once)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inference.perform_inference(model, number_iterations=500, number_samples=300, optimizer="SGD", lr=0.001)
        loss = model.diagnostics["loss curve"]
        t1 = time.perf_counter()
        calls.append(dict(wall_ms=(t1 - t0) * 1e3, final_loss=float(loss[-1]), n_losses=int(len(loss))))
    out["perform_inference_calls"] = calls
    model = W.build_readme_ar(api, T=20)
    compiled = engine.compile_model(model, None, "pathwise")
    compiled.train(500, 300, "SGD", lr=0.001)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        losses, finite = compiled.train(500, 300, "SGD", lr=0.001)
        losses.cpu()
        ts.append((time.perf_counter() - t0) * 1e3)
    out["compiled_train_500_ms"] = ts
    print("RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    cache = tempfile.mkdtemp(prefix="bsvi_cache_", dir="/tmp")
    env = dict(os.environ, BSVI_CACHE_DIR=cache)
    lines = []
    for tag in ("cold (empty code-object cache)", "cached (second process, same cache directory)"):
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        got = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
        if not got:
            lines.append("%s: FAILED\n%s" % (tag, res.stdout[-2000:]))
            continue
        d = json.loads(got[0][7:])
        c = d["perform_inference_calls"]
        lines.append("%s\n  perform_inference(model, 500, number_samples=300, optimizer='SGD', lr=0.001): first call %.1f ms, second %.1f ms, third %.1f ms "
                     "(500 iterations: %.2f / %.2f / %.2f us per iteration); final loss %.4f, %d loss entries\n  compiled.train(500, 300, 'SGD', lr=0.001) + curve to the host: %s ms"
                     % (tag, c[0]["wall_ms"], c[1]["wall_ms"], c[2]["wall_ms"], c[0]["wall_ms"] * 2, c[1]["wall_ms"] * 2, c[2]["wall_ms"] * 2,
                        c[0]["final_loss"], c[0]["n_losses"], ", ".join("%.2f" % t for t in d["compiled_train_500_ms"])))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
