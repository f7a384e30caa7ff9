"""host compile time (ingest + passes + plan + encode) of the 34-qubit circuit, no GPU needed:
    python scripts/time_compile.py            in-tree circuit container
    python scripts/time_compile.py --strict   the strict Qiskit double of tests/strict_qiskit (HAVE_QISKIT branch)
Every AND is a distinct object either way (QCMRF.py:225,227 builds a fresh one per append)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--strict" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "strict_qiskit"))
import numpy as np
import qcmrf_amd
from qcmrf_amd import QCMRF, workloads, program
from qcmrf_amd.backend import QsvBackend
from qcmrf_amd.ingest import ingest

W = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--w=")), 34))
name, C = workloads.baseline_config(4) if W == 34 else ("W=%d" % W, workloads.for_width(W))
th = workloads.theta_halfnorm(workloads.dimension(C))
t0 = time.perf_counter()
qc = QCMRF(C, th)
t_build = time.perf_counter() - t0
n_and, ids = 0, set()
for ci in qc.data:
    d = getattr(ci.operation, "definition", None) if ci.operation.name.startswith("cU_C") else None
    if d is not None:
        for c in d.data:
            if c.operation.name.startswith("and"):
                n_and += 1
                ids.add(id(c.operation))
be = QsvBackend()
best = {}
for rep in range(30):
    t0 = time.perf_counter()
    ing = ingest(qc, peephole=True)
    t1 = time.perf_counter()
    ing2, pl = be.compile(qc, 1)
    t2 = time.perf_counter()
    rec, data = program.encode(pl.ops)
    t3 = time.perf_counter()
    for k, v in (("ingest", t1 - t0), ("ingest+passes+plan", t2 - t1), ("encode", t3 - t2)):
        best[k] = min(best.get(k, 1e9), v)
print("HAVE_QISKIT=%s  %s: %d top-level instructions, %d AND instances (%d distinct objects), QCMRF() %.1f ms"
      % (qcmrf_amd.HAVE_QISKIT, name, len(qc.data), n_and, len(ids), t_build * 1e3))
print("device ops:", [o.kind for o in pl.ops][:6], "... total", len(pl.ops), " source gates", ing.n_source_ops)
print("best of 30 [ms]:", {k: round(v * 1e3, 3) for k, v in best.items()})
