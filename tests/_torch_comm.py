"""TEST INFRASTRUCTURE: torch.distributed (gloo) behind the interface of qcmrf_amd.comm -- a second
host transport for the multi-process CPU test.  The shipped package has no torch import at all
(qcmrf_amd.comm.SocketComm is its process group; tests/test_lib_abi.py enforces that)."""
import os
import sys


class TorchDistComm:
    def __init__(self, backend="gloo", init=True, timeout_s=None):
        import torch.distributed as dist
        self._dist = dist
        if init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            sys.stdout.flush()
            saved = os.dup(1)                     # gloo announces its connections with printf
            try:
                os.dup2(2, 1)
                if timeout_s:
                    import datetime
                    dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
                else:
                    dist.init_process_group(backend=backend)
            finally:
                os.dup2(saved, 1)
                os.close(saved)
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def allgather(self, obj):
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def bcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self._dist.barrier()
