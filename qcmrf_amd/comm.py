"""Host-side process group used when one process drives one GPU (torchrun-style launch).

Only tiny host values travel here (per-shard probability mass, sampled outcomes, the RCCL
unique id); amplitudes move between GPUs inside libqsv over RCCL/xGMI, never through Python.
``torch.distributed`` is imported lazily and only by ``TorchDistComm`` -- the single-GPU path
never imports torch.
"""
from __future__ import annotations

import os
import pickle
import sys


class SingleProcess:
    rank = 0
    world = 1

    def allgather(self, obj):
        return [obj]

    def bcast(self, obj, src=0):
        return obj

    def barrier(self):
        pass

    def allgather_u64(self, arr):
        return arr[None, :]


class TorchDistComm:
    """torch.distributed (gloo by default: host objects only) behind the same four calls."""

    def __init__(self, backend="gloo", init=True, timeout_s=None):
        import torch.distributed as dist
        self._dist = dist
        if init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # gloo announces its connections with printf: keep that off stdout, which belongs to
            # the caller (bench.py prints exactly one JSON line there)
            sys.stdout.flush()
            saved = os.dup(1)
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

    def allgather_u64(self, arr):
        """equal-length uint64 vectors of every rank -> (world, n) array, ONE collective (no pickling)"""
        import numpy as np
        import torch
        src = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64))
        out = torch.empty((self.world, src.numel()), dtype=torch.int64)
        try:
            self._dist.all_gather_into_tensor(out, src)
        except (RuntimeError, AttributeError):
            parts = [torch.empty_like(src) for _ in range(self.world)]
            self._dist.all_gather(parts, src)
            out = torch.stack(parts)
        return out.numpy().view(np.uint64)
