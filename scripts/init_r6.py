"""GPU box: write-only first pass with 6 register targets (one wave per SIMD) against 5 + lanes."""
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
def run(R, regs, lanes, label):
    sel = [W - 1, W - 2]
    mux = lambda t: ir.op_mux(sel, t, np.array([rx() for _ in range(4)]))
    quiet = (1 << W) - 1
    for q in regs + lanes:
        quiet &= ~(1 << q)
    ops = [ir.op_init(quiet)] + [mux(t) for t in regs + lanes]
    rec, data = program.encode(ops)
    eng.set_option("multi_r", R)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(4): eng.exec(rec, data)
    ms = eng.timer_end() / 4
    n = sum(v["launches"] for v in eng.stats()["kinds"].values()) / 4
    print("%-50s launches %.1f  %8.3f ms  %d GB/s" % (label, n, ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
R_ = lambda a, b: list(range(a, b + 1))
run(5, R_(14, 18), R_(19, 21), "R=5: reg 14-18 + borrowed 19-21")
run(6, R_(14, 19), R_(20, 21), "R=6: reg 14-19 + borrowed 20-21")
run(6, R_(6, 11), [12, 13], "R=6: reg 6-11 + borrowed 12-13")
run(5, R_(6, 10), R_(11, 13), "R=5: reg 6-10 + borrowed 11-13")
run(6, R_(6, 11), [], "R=6: reg 6-11")
run(5, R_(6, 10), [], "R=5: reg 6-10")
run(6, R_(14, 19), [], "R=6: reg 14-19")
run(6, R_(14, 19), [0, 1], "R=6: reg 14-19 + static 0-1")
