import sys, time
sys.path.insert(0, ".")
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
be = QsvBackend()
name, C = wl.baseline_config(2)
circs = [QCMRF(C, wl.theta_halfnorm(60, seed=s)) for s in range(12)]
be.run(circs[0], shots=4096, seed_simulator=1)
t0 = time.perf_counter(); [be.run(c, shots=4096, seed_simulator=1).result() for c in circs]; t1 = time.perf_counter()
r = be.run(circs, shots=4096, seed_simulator=1).result(); t2 = time.perf_counter()
print("12 x W=28 one by one: %.1f ms/circuit   as one batch (compile overlapped): %.1f ms/circuit" % ((t1 - t0) / 12 * 1e3, (t2 - t1) / 12 * 1e3))
from qcmrf_amd import run_experiment
import tempfile
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter(); run_experiment.main(["--outdir", d, "--seed-simulator", "1"]); print("run_experiment (70 circuits, 10000 shots): %.2f s" % (time.perf_counter() - t0))
