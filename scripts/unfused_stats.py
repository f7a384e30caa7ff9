import sys, json
sys.path.insert(0, ".")
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
name, C = wl.baseline_config(2)
qc = QCMRF(C, wl.theta_halfnorm(60))
be = QsvBackend()
for f in (0, 1):
    be.run(qc, shots=16, fusion=f)
    r = be.run(qc, shots=16, fusion=f, profile=True).result()
    m = r.metadata(0)
    print("fusion", f, "device ops", m["n_device_ops"], "evolve ms %.1f" % (m["time_evolve"] * 1e3))
    print(json.dumps(m["stats"], indent=0)[:900])
