"""GPU box: the generalised lane-map rule (lane_map 1: lane bit 5 on address bit 11 whenever the tile allows) against
the round-1 rule (lane_map 2: only a tile on bits 6..10) and the plain map (0): default path, full-width sweeps,
unfused stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
for label, kw in (("default path", {}), ("full-width sweeps", {"fold_fresh": False}), ("unfused stream", {"fusion": 0})):
    be = QsvBackend(**kw)
    for lm in (1, 2, 0, 1, 2):
        opts = {"lane_map": lm}
        be.run(qc, shots=16, engine_options=opts)
        r = be.run(qc, shots=16, engine_options=opts, profile=True).result()
        m = r.metadata(0)
        print("W=%d %-18s lane_map %d: evolve %8.2f ms" % (W, label, lm, m["time_evolve"] * 1e3),
              {n: (v["launches"], round(v["ms"] / v["launches"], 3), round(v["bytes"] / v["ms"] / 1e6 / 8000, 3)) for n, v in m["stats"]["kinds"].items() if v["ms"] > 0}, flush=True)
    be.close()
