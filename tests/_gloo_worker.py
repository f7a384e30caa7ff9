"""Worker of tests/test_multiproc_gloo.py: one rank per shard (CPU).  The engine is the numpy
stand-in (this container has no GPU); everything above it -- planner, program encoding, the
backend's multi-process run(), exchange orchestration, mass all-gather, multinomial shot split,
outcome merge -- is the shipped code.

argv[2] = transport of the host collectives and of the stand-in engine's shard exchange:
  gloo    torch.distributed (tests/_torch_comm.py), launched by torch.distributed.run
  socket  qcmrf_amd.comm.SocketComm -- the package's own stdlib process group -- launched as plain
          processes with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import closed_form as cf                 # noqa: E402
from oracle.sharded_numpy import NumpyEngine         # noqa: E402
from qcmrf_amd import QCMRF, workloads               # noqa: E402
from qcmrf_amd.backend import QsvBackend             # noqa: E402


def main():
    out_path = sys.argv[1]
    transport = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    if transport == "gloo":
        import torch
        import torch.distributed as dist
        from _torch_comm import TorchDistComm
        comm = TorchDistComm("gloo")

        def exchange(me, peer, send):
            t_send = torch.from_numpy(np.ascontiguousarray(send).view(np.float64).copy())
            t_recv = torch.empty_like(t_send)
            if me < peer:
                dist.send(t_send, dst=peer)
                dist.recv(t_recv, src=peer)
            else:
                dist.recv(t_recv, src=peer)
                dist.send(t_send, dst=peer)
            return t_recv.numpy().view(np.complex128)
    else:
        from qcmrf_amd.comm import SocketComm
        assert "torch" not in sys.modules
        comm = SocketComm(timeout_s=120)

        def exchange(me, peer, send):                 # every rank exchanges at the same program point
            return comm.allgather(np.ascontiguousarray(send))[peer]
    rank, world = comm.rank, comm.world

    def factory(n_qubits, devices=(0,), rank=None, world_size=None):
        assert rank == comm.rank and world_size == comm.world
        return NumpyEngine(n_qubits, world_size, owned=[rank], exchange=exchange)

    C = workloads.grid(2, 3)                          # n = 6, m = 7, W = 14
    th = workloads.theta_halfnorm(workloads.dimension(C))
    results = {}
    for layout in ("reference", "auto"):
        for fusion in (0, 2):
            be = QsvBackend(comm=comm, layout=layout, fusion=fusion)
            be._engine_factory = factory
            res = be.run(QCMRF(C, th), shots=20000, seed_simulator=11).result()
            meta = res.metadata(0)
            eng = be.last_engine
            shards = comm.allgather(eng.sh[rank])
            if rank == 0:
                amp = np.concatenate(shards)
                W = meta["n_qubits"]
                p = np.arange(2 ** W)
                l = np.zeros_like(p)
                for q, pos in enumerate(meta["layout"]):
                    l |= ((p >> pos) & 1) << q
                logical = np.empty_like(amp)
                logical[l] = amp
                err = float(np.abs(logical - cf.amplitudes(C, th)).max())
                counts = res.get_counts()
                pr = cf.probabilities(C, th)
                obs = np.zeros(2 ** W)
                for k, v in counts.items():
                    obs[int(k, 2)] += v
                sel = pr * 20000 > 5
                chi = float(((obs[sel] - pr[sel] * 20000) ** 2 / (pr[sel] * 20000)).sum() / (sel.sum() - 1))
                results["%s/%d" % (layout, fusion)] = {
                    "err": err, "n_exchanges": meta["n_exchanges"], "engine_exchanges": eng.n_exchanges,
                    "shots": int(sum(counts.values())), "outside_support": float(obs[pr == 0].sum()), "chi2": chi}
            comm.barrier()
    if rank == 0:
        json.dump(results, open(out_path, "w"))
    comm.barrier()
    if transport == "gloo":
        dist.destroy_process_group()
    else:
        assert "torch" not in sys.modules            # the package's own process group never needs it
        comm.close()


if __name__ == "__main__":
    main()
