"""GPU box: cost of lane-bit (shuffle) gates inside a k_multi pass."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(ops, label):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(6): eng.exec(rec, data)
    ms = eng.timer_end() / 6
    st = eng.stats()
    print("%-40s launches/exec %.1f  %.3f ms  %d GB/s" % (label, st["kinds"]["multi"]["launches"] / 6, ms, 32 * 2.0 ** W / ms / 1e6), flush=True)
sel = [20, 21, 22]
regs = [6, 7, 8, 9, 10]
mux = lambda t: ir.op_mux(sel, t, np.array([ru() for _ in range(8)]))
for nl in range(0, 7):
    run([mux(t) for t in regs] + [mux(t) for t in range(nl)], "5 reg + %d lane gates" % nl)
for nl in (1, 3, 6):
    run([mux(t) for t in regs[:3]] + [mux(t) for t in range(nl)], "3 reg + %d lane gates" % nl)
for b in range(6):
    run([mux(t) for t in regs] + [mux(b)], "5 reg + lane bit %d" % b)
