"""Host cost of compiling a circuit that was lowered to {cx,id,rz,sx,x} (run_experiment.py:52) at 34
qubits: ingest and passes, for the hand lowering (qcmrf_amd.transpile) and for the shape a
transpiler's clean-up passes leave (tests/_qiskit_shapes.py).  CPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _qiskit_shapes import lower_like_qiskit
from qcmrf_amd import QCMRF, workloads as wl, passes as P, ingest as I, planner, program
from qcmrf_amd.transpile import transpile
COMPACT = os.environ.get("QSV_COMPACT", "1") != "0"      # what backend.compile asks for at fusion 3
name, C = wl.baseline_config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
for label, t in (("nested (as constructed)", qc), ("hand lowering", transpile(qc)), ("transpiler-shaped", lower_like_qiskit(qc))):
    N = 5
    for _ in range(2):
        ing = I.ingest(t, peephole=True, compact=COMPACT); ops = P.optimise(ing.ops, level=3, fresh=True, flat=ing.flat)
    t0 = time.perf_counter()
    for _ in range(N): ing = I.ingest(t, peephole=True, compact=COMPACT)
    t1 = time.perf_counter()
    for _ in range(N): ops = P.optimise(ing.ops, level=3, fresh=True, flat=ing.flat)
    t2 = time.perf_counter()
    for _ in range(N):
        pl = planner.plan(ops, ing.num_qubits, 1, "auto"); program.encode(pl.ops)
    t3 = time.perf_counter()
    k = {}
    for o in ops: k[o.kind] = k.get(o.kind, 0) + 1
    print("%-24s %5d instructions: ingest %6.2f ms  passes %6.2f ms  plan+encode %5.2f ms  -> %s" % (label, len(t.data), (t1 - t0) / N * 1e3, (t2 - t1) / N * 1e3, (t3 - t2) / N * 1e3, k), flush=True)
