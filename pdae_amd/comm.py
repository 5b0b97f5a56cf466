"""Native data-parallel exchange: an RCCL communicator owned by libpdae_hip.so (include/pdae_hip.h: pdae_comm_*, pdae_allreduce_bucket),
driven with HIP streams and events instead of torch.distributed's ProcessGroupNCCL.

torch.distributed (any backend, gloo is enough) is used ONCE, as the control channel that carries the 128-byte RCCL unique id from rank 0 to
the other ranks -- the role of the TCP store in the reference's `init_process_group` (utils/utils.py:18-27).  After that the gradient buckets
of a fused step are reduced with `all_reduce(view)`: the call enqueues ncclAllReduce on this object's side stream behind an event of the
compute stream (so it starts when the bucket's last gradient kernel has finished and overlaps with the rest of the backward), and
`wait()` makes the compute stream wait for everything enqueued so far.  Opt-in: FusedRLStep(native_comm=True) / PDAE_NATIVE_RCCL=1 /
bench.py --native-rccl; the default exchange stays torch.distributed.all_reduce (async, ProcessGroupNCCL = RCCL)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import hip as H


def _torch_rccl():
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else None


class NativeComm:
    def __init__(self, device, rank=None, world=None, group=None):
        self.device = torch.device(device)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        L = H.lib()
        L.pdae_comm_unique_id.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        L.pdae_comm_init.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.pdae_allreduce_bucket.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.pdae_comm_destroy.argtypes = [ctypes.c_void_p]
        self._L, self._path = L, _torch_rccl()
        uid = (ctypes.c_char * 128)()
        if self.rank == 0:
            self._check(L.pdae_comm_unique_id(self._path, uid), "pdae_comm_unique_id")
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)          # control channel only
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0])
        torch.cuda.set_device(self.device)
        self.comm = ctypes.c_void_p()
        self._check(L.pdae_comm_init(self._path, uid, self.world, self.rank, ctypes.byref(self.comm)), "pdae_comm_init")
        self.stream = torch.cuda.Stream(device=self.device)

    def _check(self, rc, what):
        if rc != 0:
            raise H.PdaeError(f"{what} failed ({rc}): {self._L.pdae_last_error().decode()}")

    def all_reduce(self, view, op="sum"):
        """In-place all-reduce of a contiguous float32 / int32 device tensor (view): starts once the work already enqueued on the CURRENT
        stream has finished, runs on the side stream."""
        assert view.is_contiguous() and view.dtype in (torch.float32, torch.int32)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(ev)
        self._check(self._L.pdae_allreduce_bucket(self.comm, view.data_ptr(), view.numel(), int(view.dtype == torch.int32), int(op == "max"),
                                                  ctypes.c_void_p(self.stream.cuda_stream)), "pdae_allreduce_bucket")

    def wait(self):
        """The current stream waits for every all-reduce enqueued so far."""
        ev = torch.cuda.Event()
        ev.record(self.stream)
        torch.cuda.current_stream(self.device).wait_event(ev)

    def close(self):
        if self.comm:
            self._L.pdae_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()
