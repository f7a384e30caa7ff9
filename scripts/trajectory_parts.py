"""GPU box: what one branch node of a trajectory run costs (22-variable chain, 24 live qubits): the segment program alone,
behind the projection of the previous measurement (one record list), the three calls it replaces, the branch probability,
the state copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import QCMRF, _lib, ir, program, trajectory, workloads as wl
C = wl.chain(22)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
segs, width, final, nclb, cregs, nsrc = trajectory.compile_trajectory(qc)
print("live qubits", width, "segments", len(segs), flush=True)
eng = _lib.Engine(width)
other = _lib.Engine(width)
eng.exec(segs[0].rec, segs[0].data)
sg = segs[3]
slot = segs[2].measure_slot


def t(label, f, n=200):
    for _ in range(5): f()
    eng.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    eng.sync()
    print("%-52s %.3f ms" % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)


t("segment program alone (1 multiplexer)", lambda: eng.exec(sg.rec, sg.data))
t("projection + X + segment, one record list (outcome 1)", lambda: eng.exec(*sg.prog[1]))
t("projection + segment, one record list (outcome 0)", lambda: eng.exec(*sg.prog[0]))
tab = np.array([0.0, 1.0], dtype=np.complex128)
t("apply_diag (projection) alone", lambda: eng.apply_diag([slot], tab))
t("apply_mcx (X on the slot) alone", lambda: eng.apply_mcx([], slot))
t("probabilities([slot])", lambda: eng.probabilities([sg.measure_slot]))
t("copy_from", lambda: other.copy_from(eng))
for name in ("pass_budget", "single_shortcut"):
    pass
eng.close(); other.close()
