"""GPU box: tile width of the GENERAL k_multi pass (engine option general_r) on the reference's unfused
stream: per-pass time and total at W qubits."""
import sys
sys.path.insert(0, ".")
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend(fusion=0)
import os
for gr, cap in ((3, 64), (4, 64), (5, 64), (4, 96), (4, 128), (4, 256)) if not os.environ.get("QSV_CAPS_ONLY") else ((4, 64), (4, 96), (4, 128), (4, 192), (4, 256)):
    opts = {"general_r": gr, "pass_max_ops": cap}
    be.run(qc, shots=16, engine_options=opts)
    r = be.run(qc, shots=16, engine_options=opts, profile=True).result()
    m = r.metadata(0)
    print("W=%d fusion 0 general_r %d pass_max_ops %d: evolve %.2f ms" % (W, gr, cap, m["time_evolve"] * 1e3),
          {n: (v["launches"], round(v["ms"] / v["launches"], 3), round(v["bytes"] / v["ms"] / 1e6 / 8000, 3)) for n, v in m["stats"]["kinds"].items() if v["ms"] > 0}, flush=True)
