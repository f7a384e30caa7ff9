"""GPU box: one k_multi pass at R = 5 / 6 register targets with RX-like (MODE 2) or general
(MODE 1) tables and 0..6 lane gates -- what does a gate cost once the pass is paid for?"""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(ops, label):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(6): eng.exec(rec, data)
    ms = eng.timer_end() / 6
    st = eng.stats()
    print("%-44s launches/exec %.1f  %.3f ms  %d GB/s" % (label, st["kinds"]["multi"]["launches"] / 6, ms, 32 * 2.0 ** W / ms / 1e6), flush=True)
sel = [20, 21]
for R in (5, 6):
    eng.set_option("multi_r", R)
    regs = list(range(6, 6 + R))
    for kind, gen in (("rx", rx), ("general", ru)):
        mux = lambda t: ir.op_mux(sel, t, np.array([gen() for _ in range(4)]))
        for nl in (0, 1, 2, 3, 4, 6):
            run([mux(t) for t in regs] + [mux(t) for t in range(nl)], "R=%d %-7s %d reg + %d lane gates" % (R, kind, R, nl))
