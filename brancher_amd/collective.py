"""
The exchange of the multi-GPU path through the C ABI (include/bsvi.h: bsvi_allreduce, bsvi_exchange_*).

A step of the sample-sharded path is three C calls on one stream — `bsvi_elbo_fwd_bwd` (this rank's sums),
the exchange, `bsvi_finalize_step` (the replicated optimizer step).  The exchange is either RCCL's all-reduce on the
communicator torch.distributed already owns (`rccl_allreduce`), or the library's one-shot direct-write all-reduce over
HIP-IPC-mapped peer regions (`Exchange`: one one-workgroup kernel per call, for the [4 + P]-float messages of this path).
torch.distributed is used here only to hand the ranks' IPC handles around (any backend; gloo in the tests).
"""
import ctypes as C

import numpy as np
import torch

from brancher_amd import native


def rccl_comm_ptr(group=None):
    """the ncclComm_t of torch.distributed's RCCL process group (None when the backend is not nccl / not initialised)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = group or dist.distributed_c10d._get_default_group()
    try:
        backend = group._get_backend(torch.device("cuda"))
        ptr = backend._comm_ptr()
    except Exception:      # noqa: BLE001  (gloo groups, or a torch without the accessor)
        return None
    return int(ptr) or None


def rccl_allreduce(tensor, comm_ptr, stream=None):
    """in-place sum of a float32 device tensor over the ranks of the communicator (bsvi_allreduce)"""
    if tensor.dtype != torch.float32 or not tensor.is_contiguous():
        raise ValueError("bsvi_allreduce takes a contiguous float32 tensor")
    st = stream if stream is not None else torch.cuda.current_stream(tensor.device).cuda_stream
    native.check(native.load().bsvi_allreduce(C.c_void_p(comm_ptr), C.c_void_p(tensor.data_ptr()), tensor.numel(), C.c_void_p(st)))
    return tensor


class Exchange:
    """One-shot direct-write all-reduce of small float32 vectors between the ranks of one node (bsvi_exchange_*).

    Construction is collective: every rank creates its region, the IPC handles travel through torch.distributed
    (`all_gather`, any backend), every rank maps its peers' regions."""

    def __init__(self, capacity_floats, device=None, group=None):
        import torch.distributed as dist
        self.lib = native.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        handle = C.c_void_p()
        self.handle = None
        with torch.cuda.device(self.device):
            # A rank whose region cannot be created still takes part in the all_gather below (with a zero handle): were it to
            # raise first, its peers would sit in the all_gather while it went on to the next collective — mismatched, a hang.
            # Every rank sees the zero handle and every rank raises.
            rc = self.lib.bsvi_exchange_create(self.rank, self.world, int(capacity_floats), C.byref(handle))
            if rc == 0:
                self.handle = handle
            if self.world > 1:
                nbytes = int(self.lib.bsvi_exchange_handle_bytes())
                mine = (C.c_ubyte * nbytes)()
                if rc == 0:
                    rc = self.lib.bsvi_exchange_export(self.handle, mine)
                    if rc:
                        mine = (C.c_ubyte * nbytes)()
                local = torch.tensor(list(bytes(mine)), dtype=torch.uint8)
                if dist.get_backend(group) == "nccl":
                    local = local.to(self.device)
                gathered = [torch.empty_like(local) for _ in range(self.world)]
                dist.all_gather(gathered, local, group=group)
                rows = [g.cpu().numpy() for g in gathered]
                missing = [r for r, g in enumerate(rows) if not g.any()]
                if missing:
                    self.close()
                    raise native.NativeError("bsvi_exchange_create / export failed on rank(s) {}: no rank uses the exchange".format(missing))
                blob = np.concatenate(rows).astype(np.uint8)
                native.check(self.lib.bsvi_exchange_connect(self.handle, blob.ctypes.data_as(C.c_void_p)))
            else:
                native.check(rc)

    def allreduce(self, tensor, stream=None):
        if tensor.dtype != torch.float32 or not tensor.is_contiguous():
            raise ValueError("the exchange takes a contiguous float32 tensor")
        st = stream if stream is not None else torch.cuda.current_stream(tensor.device).cuda_stream
        native.check(self.lib.bsvi_exchange_allreduce(self.handle, C.c_void_p(tensor.data_ptr()), tensor.numel(), C.c_void_p(st)))
        return tensor

    def self_test(self, calls=4):
        """`calls` all-reduces of known vectors right after the regions are connected (both parities of the slot area and the
        reuse of a slot two calls later): in call k rank r contributes (r + 1 + k) * [1, 2, ...]; True when this rank got
        every exact total in time (synchronises; run once, outside any capture)"""
        n = min(64, int(getattr(self, "capacity", 64)))
        ramp = torch.arange(1, n + 1, device=self.device, dtype=torch.float32)
        ok = True
        for k in range(int(calls)):
            buf = (ramp * float(self.rank + 1 + k)).contiguous()
            self.allreduce(buf)
            torch.cuda.synchronize(self.device)
            expected = ramp * float(self.world * (self.world + 1) // 2 + k * self.world)
            ok = ok and self.status() == 0 and bool(torch.equal(buf, expected))
        # ... and the TAGGED 8-byte entry area, which only the in-kernel training loop uses (spec_main.h: spec_xput / spec_xget): the
        # same protocol as a kernel of its own (bsvi_exchange_selftest_tagged), both parities and a reuse
        st = torch.cuda.current_stream(self.device).cuda_stream
        for k in range(int(calls)):
            buf = (ramp * float(self.rank + 1 + k) + 0.25).contiguous()
            native.check(self.lib.bsvi_exchange_selftest_tagged(self.handle, C.c_void_p(buf.data_ptr()), n, C.c_void_p(st)))
            torch.cuda.synchronize(self.device)
            expected = torch.zeros_like(ramp)
            for r in range(self.world):                       # (added in rank order, as the kernel adds them)
                expected = expected + (ramp * float(r + 1 + k) + 0.25)
            ok = ok and self.status() == 0 and bool(torch.equal(buf, expected))
        return ok

    def status(self):
        """0, or the sequence number of the last call that gave up waiting for a peer (synchronises with the device)"""
        return int(self.lib.bsvi_exchange_status(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bsvi_exchange_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
