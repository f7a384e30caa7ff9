"""GPU box: where should the three targets that ride on borrowed lane bits sit?  One read+write table-op pass with
registers on bits 6..10 and borrowed targets on different address bits (non-temporal kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(regs, bor, label):
    sel = [q for q in (25, 26, 19, 21) if q not in regs + bor][:2]
    ops = [ir.op_mux(sel, t, np.array([rx() for _ in range(4)])) for t in regs + bor]
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(6): eng.exec(rec, data)
    ms = eng.timer_end() / 6
    n = sum(v["launches"] for v in eng.stats()["kinds"].values()) / 6
    print("%-34s launches %.1f  %8.3f ms  %.3f of peak" % (label, n, ms, 32 * 2.0 ** W / ms / 1e6 / 8000), flush=True)
R = list(range(6, 11))
for bor in ([], [11], [11, 12], [11, 12, 13], [13, 12, 11], [12, 13, 14], [12, 13, 11], [11, 13, 15], [14, 15, 16], [11, 12, 20], [11, 16, 20], [20, 21, 11], [11, 20, 21]):
    run(R, bor, "reg 6-10 + borrowed %s" % bor)
eng.close()
