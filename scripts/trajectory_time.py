"""GPU box: the bench's trajectory leg by itself (22-variable chain, theta scale 0.25, 4096 shots, seeds 1 / 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
tC = wl.chain(22)
tq = QCMRF(tC, wl.theta_halfnorm(wl.dimension(tC), scale=0.25))
if len(sys.argv) > 1:                                     # as in bench.py: a 34-qubit state lived on the device just before
    name, C = wl.baseline_config(4)
    big = QsvBackend()
    big.run(QCMRF(C, wl.theta_halfnorm(wl.dimension(C))), shots=64, seed_simulator=1).result()
    big.close()
    print("34-qubit state released", flush=True)
tb = QsvBackend(method="trajectory")
tb.run(tq, shots=256, seed_simulator=1)
for seed in (2, 2, 3):
    t0 = time.perf_counter()
    tr = tb.run(tq, shots=4096, seed_simulator=seed).result()
    dt = time.perf_counter() - t0
    m = tr.metadata(0)
    print("seed %d: %.0f ms, %d nodes, %d copies -> %.3f ms per node" % (seed, dt * 1e3, m["branch_nodes"], m["state_copies"], dt * 1e3 / m["branch_nodes"]), flush=True)
tb.close()
