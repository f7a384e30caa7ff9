"""Worker of tests/test_gpu_multiproc.py: one libqsv rank per PROCESS, both on device 0 (the test
box has one GPU), host-side rendezvous over the package's own process group (qcmrf_amd.comm).  RCCL refuses two ranks on one device, so the
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
from qcmrf_amd.comm import SocketComm                # noqa: E402


def config4(comm, out_path):
    """BASELINE config 4 (W = 31, 32 GiB) sharded over the ranks: 16 GiB (2 ranks) / 8 GiB (4 ranks)
    shards, reference layout with full-width sweeps so that the planner must exchange half-shards
    (peer-mapped transport here: the ranks share the box's one GPU).  Too wide to gather: every
    rank checks slices of ITS shard against the closed form; masses and counts are merged."""
    rank, world = comm.rank, comm.world
    name, C = workloads.baseline_config(3)
    n, m, W, dim = cf.model_shape(C)
    th = workloads.theta_halfnorm(dim)
    L = W - (world.bit_length() - 1)
    results = {}
    for layout, fold in (("reference", False), ("auto", False), ("auto", True)):
        be = QsvBackend(comm=comm, device=0, layout=layout, fold_fresh=fold)
        res = be.run(QCMRF(C, th), shots=4096, seed_simulator=1984).result()
        meta = res.metadata(0)
        eng = be.last_engine
        lay = meta["layout"]
        rs = np.random.RandomState(rank)
        err = 0.0
        starts = [0, (1 << L) - 4096] + [int(x) for x in rs.randint(0, (1 << L) - 4096, size=8)]
        for st0 in starts:
            g0 = (rank << L) + st0
            got = eng.amplitudes(g0, 4096)
            pidx = np.arange(g0, g0 + 4096, dtype=np.uint64)
            lidx = np.zeros_like(pidx)
            for q, pos in enumerate(lay):
                lidx |= ((pidx >> np.uint64(pos)) & np.uint64(1)) << np.uint64(q)
            err = max(err, float(np.abs(got - cf.amplitudes_at(C, th, lidx)).max()))
        errs = comm.allgather(err)
        masses = comm.allgather(eng.norm())
        st = eng.stats()
        xs = comm.allgather(int(st["exchanges"]))
        counts = res.get_counts()
        if rank == 0:
            p, Z = cf.gibbs_pmf(C, th)
            ok = sum(v for k, v in counts.items() if int(k, 2) < 2 ** n)
            pr = cf.probabilities_at(C, th, np.array([int(k, 2) for k in counts], dtype=np.uint64))
            results["%s/%s" % (layout, "fold" if fold else "sweeps")] = {
                "err": max(errs), "norm": float(sum(masses)), "n_exchanges": meta["n_exchanges"], "engine_exchanges": xs,
                "transport": getattr(eng, "transport", None), "shots": int(sum(counts.values())),
                "outside_support": int((pr <= 0).sum()), "success": ok / 4096.0, "delta": float(Z / 2 ** n),
                "evolve_ms": meta["time_evolve"] * 1e3}
        comm.barrier()
        be.close()
    if rank == 0:
        json.dump(results, open(out_path, "w"))
    comm.barrier()


def main():
    out_path = sys.argv[1]
    comm = SocketComm(timeout_s=300)
    rank, world = comm.rank, comm.world
    if len(sys.argv) > 2 and sys.argv[2] == "config4":
        config4(comm, out_path)
        comm.close()
        return
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
    assert "torch" not in sys.modules                # launched by torch.distributed.run, never imported here
    comm.close()


if __name__ == "__main__":
    main()
