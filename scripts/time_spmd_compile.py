"""Host cost of compiling one 34-qubit QCMRF circuit on N rank processes, every rank for itself (the default) and SPMD
(each rank reads 1/N of the clique blocks, ONE all-gather completes the program: ingest(..., comm=)).  No GPU: the
ranks only compile, in lockstep (a barrier before every compile), which is what ranks with a GPU each do -- the
one-GPU rehearsal cannot show it (its ranks wait for each other's kernels inside the all-gather).
  usage: python scripts/time_spmd_compile.py [N ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main():
    from qcmrf_amd import QCMRF, workloads as wl
    from qcmrf_amd.backend import QsvBackend
    from qcmrf_amd.comm import SocketComm
    comm = SocketComm(timeout_s=60)
    name, C = wl.baseline_config(4)
    qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
    be = QsvBackend()
    rows = {}
    for label, opts in (("every rank for itself", {}), ("SPMD", {"ingest_comm": comm})):
        ts = []
        for i in range(40):
            comm.barrier()
            t0 = time.perf_counter()
            be.compile(qc, **opts)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[5:])
        rows[label] = (ts[len(ts) // 2], ts[0])
    allrows = comm.allgather(rows)
    if comm.rank == 0:
        for label in rows:
            print("  %-22s median ms per rank: %s   (best: %s)" % (label, " ".join("%.2f" % r[label][0] for r in allrows),
                                                                 " ".join("%.2f" % r[label][1] for r in allrows)), flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    if os.environ.get("QSV_SPMD_CHILD"):
        rank_main()
        sys.exit(0)
    for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        print("%d ranks on %d cores" % (n, os.cpu_count()), flush=True)
        ep = "unix:/tmp/qsv-spmd-%d-%d" % (os.getpid(), n)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)],
                                  env=dict(os.environ, QSV_SPMD_CHILD="1", RANK=str(r), WORLD_SIZE=str(n), QSV_COMM_ENDPOINT=ep))
                 for r in range(n)]
        rc = [p.wait(timeout=300) for p in procs]
        if any(rc):
            sys.exit("a rank failed: %r" % rc)
