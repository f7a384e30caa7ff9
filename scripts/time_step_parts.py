"""GPU box: one step split into compile / exec / sample (C) / post-processing (Python), per config."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import QCMRF, workloads as wl, program
from qcmrf_amd.backend import QsvBackend, _format_keys

for ci in [int(a) for a in sys.argv[1:]] or [2, 4]:
    name, C = wl.baseline_config(ci)
    qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
    be = QsvBackend()
    be.run(qc, shots=4096, seed_simulator=1).result()
    eng = be.last_engine
    W = qc.num_qubits
    ing, pl = be.compile(qc)
    rec, data = program.encode(pl.ops)
    clist = sorted(ing.measure)
    meas = [pl.layout[ing.measure[c]] for c in clist]
    tt = {"compile": 0, "exec": 0, "sample_c": 0, "post": 0}
    n = 10
    for i in range(n + 2):
        t0 = time.perf_counter()
        be.compile(qc); program.encode(pl.ops)
        t1 = time.perf_counter()
        eng.exec(rec, data); eng.sync()
        t2 = time.perf_counter()
        bits = eng.sample(4096, 7 + i, meas)
        t3 = time.perf_counter()
        vals = np.zeros(bits.shape, dtype=np.uint64)
        for j, c in enumerate(clist):
            vals |= ((bits >> np.uint64(j)) & np.uint64(1)) << np.uint64(c)
        uv, uc = np.unique(vals, return_counts=True)
        counts = _format_keys(uv, uc, ing.num_clbits, ing.creg_sizes)
        t4 = time.perf_counter()
        if i >= 2:
            for k, v in zip(tt, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                tt[k] += v / n * 1e3
    print(name[:8], {k: round(v, 3) for k, v in tt.items()}, "keys", len(counts), flush=True)
    be.close()
