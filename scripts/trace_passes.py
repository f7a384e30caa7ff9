"""GPU box: one line per k_multi pass of the reference's unfused stream (engine option trace_passes):
register count, kernel mode, ops by update shape -- then the same run timed per kernel."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
fusion = int(sys.argv[2]) if len(sys.argv) > 2 else 0
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend(fusion=fusion)
be.run(qc, shots=16)
be.run(qc, shots=16, engine_options={"trace_passes": 1})
r = be.run(qc, shots=16, profile=True).result()
m = r.metadata(0)
print("W=%d fusion %d: evolve %.2f ms" % (W, fusion, m["time_evolve"] * 1e3),
      {n: (v["launches"], round(v["ms"] / v["launches"], 3)) for n, v in m["stats"]["kinds"].items() if v["ms"] > 0})
