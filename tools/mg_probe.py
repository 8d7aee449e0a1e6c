"""Diagnostic: host and device cost per iteration of the multi-GPU step sequence, run on one GPU
(bsvi_elbo_fwd_bwd -> all_reduce (world size 1) -> bsvi_finalize_step)."""
import os, time, ctypes as C, torch, torch.distributed as dist, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29513", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from brancher_amd import config, engine, native, workloads as W
config.set_device("cuda:0")
c = engine.compile_model(W.build_readme_ar(W.native_api(), T=20), None, "pathwise")
p = c.program
cfg = native.make_opt_cfg("SGD", lr=1e-3)
state = torch.zeros(4 * p.n_params, device="cuda")
loss_curve = torch.zeros(2000, device="cuda"); finite = torch.ones(2000, device="cuda")
ptr = lambda t: C.c_void_p(t.data_ptr())
def run(K, with_ar):
    for it in range(K):
        args = c._elbo_args(300, 300, 0, None, 0, it)
        native.check(c.lib.bsvi_elbo_fwd_bwd(c.native.handle, C.byref(args)))
        if with_ar: dist.all_reduce(c.out)
        native.check(c.lib.bsvi_finalize_step(C.byref(cfg), ptr(c.params), ptr(c.out), ptr(state), ptr(c.mask_all), p.n_params, 300,
                                              C.c_void_p(loss_curve.data_ptr() + 4 * it), C.c_void_p(finite.data_ptr() + 4 * it), c._stream()))
for with_ar in (False, True):
    run(100, with_ar); torch.cuda.synchronize(); t = time.time(); run(1000, with_ar); t1 = time.time() - t; torch.cuda.synchronize(); t2 = time.time() - t
    print("allreduce" if with_ar else "no allreduce", "host us/iter %.1f  total us/iter %.1f" % (t1 * 1e3, t2 * 1e3))
dist.destroy_process_group()
