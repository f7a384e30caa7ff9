"""GPU box: the reference's experiment itself (run_experiment.py:20-57: 7 graphs x 10 theta draws = 70 circuits of 4-12
qubits, 10 000 shots each) through run().result().get_counts() -- as constructed, and lowered to {cx,id,rz,sx,x} first as
run_experiment.py:52 does (stand-in transpiler, not timed).  These circuits are microseconds of device work: the time is host
compile + launches + sampling, with the compile of circuit i + 1 overlapped with the device work of circuit i."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.stats import halfnorm
from qcmrf_amd import QCMRF, Aer
from qcmrf_amd.transpile import transpile
from qcmrf_amd.workloads import REFERENCE_GRAPHS as GRAPHS

np.random.seed(1984)
circs = []
for j, C in enumerate(GRAPHS):
    d = sum(2 ** len(c) for c in C)
    for _ in range(10):
        circs.append(QCMRF(C, (-halfnorm.rvs(loc=0, scale=0.5, size=d)).tolist(), with_measurements=True))
low = [transpile(c) for c in circs]
sim = Aer.get_backend("qasm_simulator")
sim.run(circs[:2], shots=100).result()
for label, cs in (("as constructed", circs), ("lowered to basis gates", low)):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        counts = sim.run(cs, shots=10000, seed_simulator=rep).result().get_counts()
        best = min(best, time.perf_counter() - t0)
    assert len(counts) == 70 and all(sum(c.values()) == 10000 for c in counts)
    print("%-24s 70 circuits x 10000 shots: %.1f ms in run().result().get_counts() = %.2f ms per circuit (%d-%d gates each)"
          % (label, best * 1e3, best / 70 * 1e3, min(len(c.data) for c in cs), max(len(c.data) for c in cs)), flush=True)
