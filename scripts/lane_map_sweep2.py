"""GPU box: register-bit placement x lane map, one k_multi pass of 5 RX-like gates (r+w, init)."""
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
    return (16 if init else 32) * 2.0 ** W / ms / 1e6
R = lambda a, b: list(range(a, b + 1))
for init in (False, True):
    print("init" if init else "r+w", "GB/s; columns: lane map", flush=True)
    for a in range(6, 15):
        regs = R(a, a + 4)
        hi = a + 4
        maps = [(0, 0, 0), (0, 0, hi + 1), (0, 0, hi + 2), (0, hi + 1, hi + 2), (hi + 1, hi + 2, hi + 3)]
        if a > 6: maps += [(0, 0, a - 1)]
        if a > 7: maps += [(0, a - 2, a - 1)]
        if a > 8: maps += [(a - 3, a - 2, a - 1)]
        out = []
        for lm in maps:
            out.append("%s:%d" % (",".join(str(x) for x in lm), run(regs, lm, init)))
        print("reg %2d-%2d  " % (a, hi) + "  ".join(out), flush=True)
