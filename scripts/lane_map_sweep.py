"""GPU box: the lane map as a pure access-pattern knob.  One k_multi pass (5 register targets),
lane bits 3..5 mapped to various free address bits; read+write and init-fused; W qubits."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(regs, lm, init):
    sel = [W - 1, W - 2]
    mux = lambda t: ir.op_mux(sel, t, np.array([rx() for _ in range(4)]))
    quiet = (1 << W) - 1
    for q in regs:
        quiet &= ~(1 << q)
    ops = ([ir.op_init(quiet)] if init else []) + [mux(t) for t in regs]
    rec, data = program.encode(ops)
    eng.set_option("lane_map", lm[0] | lm[1] << 5 | lm[2] << 10)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.timer_begin()
    for _ in range(4): eng.exec(rec, data)
    ms = eng.timer_end() / 4
    print("%-4s reg %-12s lanes3-5 -> %-14s %8.3f ms  %d GB/s" % ("init" if init else "r+w", "%d-%d" % (regs[0], regs[-1]), lm, ms, (16 if init else 32) * 2.0 ** W / ms / 1e6), flush=True)
R = lambda a, b: list(range(a, b + 1))
for init in (False, True):
    for regs in (R(6, 10), R(14, 18)):
        hi = regs[-1]
        lo_free = 6 if regs[0] > 8 else hi + 1
        maps = [(0, 0, 0), (0, 0, hi + 1), (0, 0, hi + 2), (0, 0, hi + 3), (0, 0, hi + 5), (0, hi + 1, hi + 2), (0, hi + 2, hi + 3),
                (hi + 1, hi + 2, hi + 3), (0, 0, lo_free), (0, lo_free, lo_free + 1), (0, 0, 22), (0, 0, 26)]
        for lm in maps:
            run(regs, lm, init)
