"""Worker of tests/test_gpu_multiproc.py: one libqsv rank per PROCESS, both on device 0 (the test
box has one GPU), host-side rendezvous over gloo.  RCCL refuses two ranks on one device, so the
shard-bit exchange goes over the peer-mapped (HIP IPC) transport; everything else -- planner swap
insertion, qsv_create_rank, per-shard gate resolution, the one-collective sampling merge -- is
the path a real multi-GPU launch takes."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from qcmrf_amd import _lib                           # noqa: E402  (HIP runtime of /opt/rocm first)
_lib.load()
assert _lib.device_count() >= 1

from oracle import closed_form as cf                 # noqa: E402
from qcmrf_amd import QCMRF, workloads               # noqa: E402
from qcmrf_amd.backend import QsvBackend             # noqa: E402
from qcmrf_amd.comm import TorchDistComm             # noqa: E402


def main():
    out_path = sys.argv[1]
    comm = TorchDistComm("gloo")
    rank, world = comm.rank, comm.world
    C = workloads.chain(8)                            # n = 8, m = 7, W = 16: L = 15 / 14 local qubits
    th = workloads.theta_halfnorm(workloads.dimension(C))
    want = cf.amplitudes(C, th)
    pr = np.abs(want) ** 2
    results = {}
    for layout, fusion, fold in (("reference", 2, False), ("reference", 0, False), ("auto", 3, True), ("auto", 2, False)):
        be = QsvBackend(comm=comm, device=0, layout=layout, fusion=fusion, fold_fresh=fold)
        res = be.run(QCMRF(C, th), shots=20000, seed_simulator=5).result()
        meta = res.metadata(0)
        eng = be.last_engine
        W = meta["n_qubits"]
        L = W - (world.bit_length() - 1)
        mine = eng.amplitudes(rank << L, 1 << L)
        shards = comm.allgather(mine)
        st = eng.stats()
        xs = comm.allgather(int(st["exchanges"]))
        if rank == 0:
            amp = np.concatenate(shards)
            p = np.arange(2 ** W)
            l = np.zeros_like(p)
            for q, pos in enumerate(meta["layout"]):
                l |= ((p >> pos) & 1) << q
            logical = np.empty_like(amp)
            logical[l] = amp
            counts = res.get_counts()
            obs = np.zeros(2 ** W)
            for k, v in counts.items():
                obs[int(k, 2)] += v
            sel = pr * 20000 > 5
            chi = float(((obs[sel] - pr[sel] * 20000) ** 2 / (pr[sel] * 20000)).sum() / (sel.sum() - 1))
            results["%s/%d" % (layout, fusion)] = {
                "err": float(np.abs(logical - want).max()), "n_exchanges": meta["n_exchanges"], "engine_exchanges": xs,
                "transport": getattr(eng, "transport", None), "shots": int(sum(counts.values())),
                "outside_support": float(obs[pr == 0].sum()), "chi2": chi}
        comm.barrier()
        be.close()
    if rank == 0:
        json.dump(results, open(out_path, "w"))
    comm.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
