"""GPU box: 34-qubit full-width sweeps (fold_fresh off) with the LAST pass's register targets on the top bits of
the shard (planner default from 2^33 amplitudes) against the low bits, with the non-temporal kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl, planner
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend(fold_fresh=False)
for label, thr in (("top (default)", 33), ("low", 99), ("top (default)", 33), ("low", 99)):
    planner.GEN_TOP_MIN_L = thr
    for nt in (-1, 0):
        opts = {"multi_nt": nt}
        be.run(qc, shots=16, engine_options=opts)
        r = be.run(qc, shots=16, engine_options=opts, profile=True).result()
        m = r.metadata(0)
        print("W=%d last pass %-14s multi_nt %2d: evolve %.2f ms" % (W, label, nt, m["time_evolve"] * 1e3),
              {n: (v["launches"], round(v["ms"] / v["launches"], 3), round(v["bytes"] / v["ms"] / 1e6 / 8000, 3)) for n, v in m["stats"]["kinds"].items() if v["ms"] > 0}, flush=True)
