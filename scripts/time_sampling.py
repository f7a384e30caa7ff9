"""GPU box: where does the sampling half of a step go?  (run from the repo root)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend, _format_keys

name, C = wl.baseline_config(2)
qc = QCMRF(C, wl.theta_halfnorm(60))
be = QsvBackend()
be.run(qc, shots=4096, seed_simulator=1).result()
eng = be.last_engine
W = qc.num_qubits
meas = list(range(W))
def t(f, n=20):
    f(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
print("norm            %.3f ms" % t(lambda: eng.norm())[0])
ms, bits = t(lambda: eng.sample(4096, 7, meas))
print("sample(4096)    %.3f ms" % ms)
print("sample(64)      %.3f ms" % t(lambda: eng.sample(64, 7, meas))[0])
print("sample(65536)   %.3f ms" % t(lambda: eng.sample(65536, 7, meas))[0])
ms, (uv, uc) = t(lambda: np.unique(bits, return_counts=True))
print("np.unique       %.3f ms (%d keys)" % (ms, len(uv)))
print("format keys     %.3f ms" % t(lambda: _format_keys(uv, uc, W, None))[0])
for fusion in (2, 0):
    ms, _ = t(lambda: be.compile(qc, fusion=fusion), 5)
    print("compile fusion=%d %.3f ms" % (fusion, ms))
ms, _ = t(lambda: be.run(qc, shots=4096, seed_simulator=3).result(), 10)
print("full step       %.3f ms" % ms)
