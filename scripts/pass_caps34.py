"""GPU box: ops per general pass (pass_max_ops) against total time and per-pass fraction of the stream, unfused 34-qubit
reference stream (fusion 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend()
for cap in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (40, 48, 56, 64, 80, 96, 112)):
    eo = {"pass_max_ops": cap}
    be.run(qc, shots=16, fusion=0, engine_options=eo)
    r = be.run(qc, shots=16, fusion=0, profile=True, engine_options=eo).result()
    k = r.metadata(0)["stats"]["kinds"]
    print("pass_max_ops %3d: evolve %7.1f ms | " % (cap, r.metadata(0)["time_evolve"] * 1e3) +
          "  ".join("%s: %d x %.2f ms (%.3f)" % (n, v["launches"], v["ms"] / v["launches"], v["bytes"] / v["ms"] / 1e6 / 8000) for n, v in k.items() if v["launches"]), flush=True)
be.close()
